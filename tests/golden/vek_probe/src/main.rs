// Prints the bit patterns vek 0.17.2 produces for the four operations whose rounding the MI355X back end had to restate
// without the crate's source (include/rusterix_vek.hpp): Mat4 * Vec4, Mat4 * Mat4, Vec3::normalized, Vec3::lerp, and the two
// reductions the lighting code calls directly, Vec3::dot and Vec3::magnitude.
// Build and run where cargo exists, then:  cargo run --release > probe.txt && python3 compare.py probe.txt
// Inputs come from an integer LCG turned into floats by exact operations, identical in gen_expected.cpp.
use vek::{Mat4, Vec3, Vec4};

fn lcg(s: &mut u32) -> f32 {
    *s = s.wrapping_mul(1664525).wrapping_add(1013904223);
    ((*s >> 8) as f32) / 16777216.0 * 8.0 - 4.0
}
fn mat(s: &mut u32) -> Mat4<f32> {
    let mut c = [[0f32; 4]; 4];
    for col in c.iter_mut() {
        for x in col.iter_mut() {
            *x = lcg(s);
        }
    }
    Mat4::from_col_arrays(c)
}
fn hex(v: &[f32]) -> String {
    v.iter().map(|x| format!("{:08x}", x.to_bits())).collect::<Vec<_>>().join(" ")
}

fn main() {
    let mut s = 0x52585231u32;
    for i in 0..64 {
        let m = mat(&mut s);
        let n = mat(&mut s);
        let v = Vec4::new(lcg(&mut s), lcg(&mut s), lcg(&mut s), lcg(&mut s));
        let a = Vec3::new(lcg(&mut s), lcg(&mut s), lcg(&mut s));
        let b = Vec3::new(lcg(&mut s), lcg(&mut s), lcg(&mut s));
        let t = lcg(&mut s) / 8.0 + 0.5;
        let mv = m * v;
        let mm = (m * n).into_col_arrays();
        let nv = a.normalized();
        let l = Vec3::lerp(a, b, t);
        println!("{} matvec {}", i, hex(&[mv.x, mv.y, mv.z, mv.w]));
        println!("{} matmat {}", i, hex(&mm.concat()));
        println!("{} normalized {}", i, hex(&[nv.x, nv.y, nv.z]));
        println!("{} lerp {}", i, hex(&[l.x, l.y, l.z]));
        println!("{} dot {}", i, hex(&[a.dot(b)]));
        println!("{} magnitude {}", i, hex(&[a.magnitude()]));
    }
}
