//! `RasterizeHip`: drop-in for `rusterix::Rasterizer::rasterize` that keeps scene set-up, `Scene::project` and the `Edges`
//! precompute in Rust (reference src/rasterizer.rs:185-223) and hands everything after it (`:256-579`) to the gfx950 kernels
//! through include/rxr.h (ABI 4; `ffi.rs` is generated from the header by tools/gen_ffi.py and asserts every struct layout at
//! compile time).
//!
//! Usage in examples/cube.rs, examples/obj.rs, examples/map.rs -- one changed line:
//!
//! ```ignore
//! use rusterix_hip_shim::RasterizeHip;            // + this import
//! Rasterizer::setup(None, view, proj)
//!     .ambient(Vec4::one())
//!     .rasterize_hip(&mut scene, pixels, width, height, 40, &assets);   // was .rasterize(..)
//! ```
//!
//! What crosses the boundary per frame: the projected batches of every list in submission order (chunks: opacity, opaque,
//! terrain; then static, dynamic, overlay), lights, occluders, linedefs, the per-chunk data (`rxr_chunk`: occluders, program
//! range, baked shader textures, terrain texture / origin / size).  Once per change: textures (`rxr_set_textures`, entity /
//! item sequence tiles resolved here and appended to the dynamic tiles) and Rusteria programs (`rxr_set_shaders`: the `NodeOp`
//! trees serialised depth-first by `serialise_ops`, pattern banks, palette).  With `RXR_DEVICE_PROJECTION=1` the object-space
//! batches are registered once (`rxr_set_meshes`) and a frame sends matrices only.  `RXR_DEVICES=0,1,2,3` makes the context a
//! multi-device one (`rxr_create_multi`): the library shards every frame over those GPUs by itself.
//!
//! CPU fallback (HERE, in the caller's crate -- the library itself never falls back): no GPU (`rxr_create` fails), a render
//! graph with nodes (the editor's procedural sky: `render_miss_d3` / `render_setup`, src/rasterizer.rs:227-253, :424-432, is not
//! on the device), or any error status of the library -- `RXR_ERR_UNSUPPORTED` for programs whose result depends on the
//! reference's per-tile `Execution` state (impure programs, leaking `SetEmissive`, a fourth nested opacity batch, ...).
//!
//! NOT compiled in this repository's build image (no rustc / cargo there).  It is written against the reference snapshot's
//! types (file:line cited at each use) plus the accessor patch of INTEGRATION.md section 2.
pub mod ffi;

use ffi::*;
use rusteria::{NodeOp, Program};
use rusterix::prelude::*;
use std::sync::Mutex;
use vek::Mat4;

// ---- context ---------------------------------------------------------------------------------------------------
struct Ctx(*mut rxr_ctx);
unsafe impl Send for Ctx {}

#[derive(Default)]
struct Caches {
    ctx: Option<Ctx>,
    tried: bool,
    textures: u64, // fingerprints of what the device currently holds
    shaders: u64,
    meshes: u64,
    meshes2d: u64,
    /// streaming hand-over (project_streaming): the repacked arrays of the frame's 3D batches, in submission order, as the device
    /// was given them -- device_frame must name exactly these pointers again
    stream: Vec<Repack>,
    streamed: bool,
}
static STATE: Mutex<Option<Caches>> = Mutex::new(None);

fn create_context() -> Option<Ctx> {
    let mut p: *mut rxr_ctx = std::ptr::null_mut();
    // RXR_DEVICES=0,1,2,3: one member context per GPU, frames sharded inside the library (include/rxr.h rxr_create_multi)
    if let Ok(list) = std::env::var("RXR_DEVICES") {
        let ids: Vec<i32> = list.split(',').filter_map(|s| s.trim().parse().ok()).collect();
        if ids.len() > 1 {
            return if unsafe { rxr_create_multi(&mut p, ids.as_ptr(), ids.len() as i32) } == RXR_OK { Some(Ctx(p)) } else { None };
        }
    }
    let dev = std::env::var("RXR_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
    if unsafe { rxr_create(&mut p, dev) } == RXR_OK { Some(Ctx(p)) } else { None }
}

/// Arithmetic of the 3D light loop on the device (include/rxr.h `rxr_set_light_math`): `exact = true` reproduces the CPU
/// rasterizer's lit frames up to libm's log2f / exp2f; `false` (the library's default) stays within 1 per 8-bit channel and
/// renders lit frames a third faster.  Applies from the next frame on; `RXR_LIGHT_MATH=exact|relaxed` in the environment
/// overrides it.  Returns false when there is no device context (the CPU path is exact anyway).
pub fn set_light_math(exact: bool) -> bool {
    let mut guard = STATE.lock().unwrap();
    let st = guard.get_or_insert_with(Caches::default);
    if st.ctx.is_none() && !st.tried {
        st.tried = true;
        st.ctx = create_context();
    }
    match &st.ctx {
        Some(ctx) => unsafe { rxr_set_light_math(ctx.0, if exact { RXR_LIGHT_MATH_EXACT as i32 } else { RXR_LIGHT_MATH_RELAXED as i32 }) == RXR_OK },
        None => false,
    }
}

fn fnv(h: &mut u64, bytes: &[u8]) {
    for b in bytes {
        *h = (*h ^ *b as u64).wrapping_mul(1099511628211);
    }
}
fn fnv_usize(h: &mut u64, v: usize) {
    fnv(h, &(v as u64).to_le_bytes());
}
/// Content fingerprint of a buffer that is expected to stay put (static tile textures, registered meshes): its length plus 256
/// evenly spaced 16-byte probes and both ends.  Cheap enough for every frame (4 KB read per buffer) and catches the edits the
/// reference allows in place (`Texture::set_pixel` / `fill` on `pub data`, a reallocation at the same address): a stale texel on
/// the GPU would otherwise render silently wrong, where the reference reads texture memory every frame.  A change that touches
/// none of the probes is not seen -- callers that edit a resident buffer in place call `invalidate_device_caches()`.
fn fnv_sampled(h: &mut u64, bytes: &[u8]) {
    fnv_usize(h, bytes.len());
    if bytes.len() <= 8192 {
        fnv(h, bytes);
        return;
    }
    let step = bytes.len() / 256;
    for k in 0..256 {
        let at = k * step;
        fnv(h, &bytes[at..at + 16]);
    }
    fnv(h, &bytes[bytes.len() - 64..]);
}
fn as_bytes<T>(v: &[T]) -> &[u8] {
    // (plain-old-data slices only: f32 / usize tuples / Vec3)
    unsafe { std::slice::from_raw_parts(v.as_ptr() as *const u8, std::mem::size_of_val(v)) }
}

/// Forgets what the shim believes the device holds (textures, programs, registered meshes): the next frame uploads all of it
/// again.  For callers that edit a resident texture or mesh in place in a way the sampled fingerprints cannot see.
pub fn invalidate_device_caches() {
    if let Some(st) = STATE.lock().unwrap().as_mut() {
        st.textures = 0;
        st.shaders = 0;
        st.meshes = 0;
        st.meshes2d = 0;
    }
}

// ---- NodeOp tree -> the word stream of include/rxr.h ------------------------------------------------------------
/// Depth-first serialisation of `Program.user_functions[i]` (rusteria/src/node/nodeop.rs:12-103): opcode = the variant's
/// position in the enum, payloads as documented in include/rxr.h.  Block lengths are in words.
fn serialise_ops(ops: &[NodeOp], out: &mut Vec<u32>, uses_patterns: &mut bool) {
    for op in ops {
        match op {
            NodeOp::LoadGlobal(i) => out.extend_from_slice(&[RXR_NODE_LOAD_GLOBAL, *i as u32]),
            NodeOp::StoreGlobal(i) => out.extend_from_slice(&[RXR_NODE_STORE_GLOBAL, *i as u32]),
            NodeOp::LoadLocal(i) => out.extend_from_slice(&[RXR_NODE_LOAD_LOCAL, *i as u32]),
            NodeOp::StoreLocal(i) => out.extend_from_slice(&[RXR_NODE_STORE_LOCAL, *i as u32]),
            NodeOp::GetComponents(sw) | NodeOp::SetComponents(sw) => {
                out.push(if matches!(op, NodeOp::GetComponents(_)) { RXR_NODE_GET_COMPONENTS } else { RXR_NODE_SET_COMPONENTS });
                out.push(sw.len() as u32);
                out.extend(sw.iter().map(|c| *c as u32));
            }
            NodeOp::If(then_code, else_code) => {
                let (mut t, mut e) = (vec![], vec![]);
                serialise_ops(then_code, &mut t, uses_patterns);
                if let Some(ec) = else_code {
                    serialise_ops(ec, &mut e, uses_patterns);
                }
                out.extend_from_slice(&[RXR_NODE_IF, t.len() as u32, else_code.is_some() as u32, e.len() as u32]);
                out.extend(t);
                out.extend(e);
            }
            // For(init, cond, incr, body): rusteria/src/node/execution.rs:247
            NodeOp::For(init, cond, incr, body) => {
                let mut blocks: [Vec<u32>; 4] = Default::default();
                for (dst, src) in blocks.iter_mut().zip([init, cond, incr, body]) {
                    serialise_ops(src, dst, uses_patterns);
                }
                out.push(RXR_NODE_FOR);
                out.extend(blocks.iter().map(|b| b.len() as u32));
                for b in blocks {
                    out.extend(b);
                }
            }
            NodeOp::Push(v) => out.extend_from_slice(&[RXR_NODE_PUSH, v.x.to_bits(), v.y.to_bits(), v.z.to_bits()]),
            NodeOp::FunctionCall(arity, total_locals, index) => {
                out.extend_from_slice(&[RXR_NODE_FUNCTION_CALL, *arity as u32, *total_locals as u32, *index as u32])
            }
            NodeOp::Sample => {
                *uses_patterns = true;
                out.push(RXR_NODE_SAMPLE)
            }
            NodeOp::SampleNormal => {
                *uses_patterns = true;
                out.push(RXR_NODE_SAMPLE_NORMAL)
            }
            other => out.push(unit_opcode(other)),
        }
    }
}

/// the payload-free variants, in declaration order
fn unit_opcode(op: &NodeOp) -> u32 {
    use NodeOp::*;
    match op {
        Swap => RXR_NODE_SWAP, Return => RXR_NODE_RETURN, Dup => RXR_NODE_DUP, Clear => RXR_NODE_CLEAR, Pack2 => RXR_NODE_PACK2,
        Pack3 => RXR_NODE_PACK3, Add => RXR_NODE_ADD, Sub => RXR_NODE_SUB, Mul => RXR_NODE_MUL, Div => RXR_NODE_DIV,
        Length => RXR_NODE_LENGTH, Length2 => RXR_NODE_LENGTH2, Length3 => RXR_NODE_LENGTH3, Abs => RXR_NODE_ABS, Sin => RXR_NODE_SIN,
        Sin1 => RXR_NODE_SIN1, Sin2 => RXR_NODE_SIN2, Cos => RXR_NODE_COS, Cos1 => RXR_NODE_COS1, Cos2 => RXR_NODE_COS2,
        Tan => RXR_NODE_TAN, Atan => RXR_NODE_ATAN, Atan2 => RXR_NODE_ATAN2, Rotate2D => RXR_NODE_ROTATE2D, Dot => RXR_NODE_DOT,
        Dot2 => RXR_NODE_DOT2, Dot3 => RXR_NODE_DOT3, Cross => RXR_NODE_CROSS, Normalize => RXR_NODE_NORMALIZE, Floor => RXR_NODE_FLOOR,
        Ceil => RXR_NODE_CEIL, Round => RXR_NODE_ROUND, Fract => RXR_NODE_FRACT, Mod => RXR_NODE_MOD, Degrees => RXR_NODE_DEGREES,
        Radians => RXR_NODE_RADIANS, Min => RXR_NODE_MIN, Max => RXR_NODE_MAX, Mix => RXR_NODE_MIX, Smoothstep => RXR_NODE_SMOOTHSTEP,
        Step => RXR_NODE_STEP, Clamp => RXR_NODE_CLAMP, Sqrt => RXR_NODE_SQRT, Pow => RXR_NODE_POW, Log => RXR_NODE_LOG,
        Print => RXR_NODE_PRINT, Eq => RXR_NODE_EQ, Ne => RXR_NODE_NE, Lt => RXR_NODE_LT, Le => RXR_NODE_LE, Gt => RXR_NODE_GT,
        Ge => RXR_NODE_GE, And => RXR_NODE_AND, Or => RXR_NODE_OR, Not => RXR_NODE_NOT, Neg => RXR_NODE_NEG, UV => RXR_NODE_UV,
        SetUV => RXR_NODE_SET_UV, Normal => RXR_NODE_NORMAL, SetNormal => RXR_NODE_SET_NORMAL, Hitpoint => RXR_NODE_HITPOINT,
        Time => RXR_NODE_TIME, Color => RXR_NODE_COLOR, SetColor => RXR_NODE_SET_COLOR, Roughness => RXR_NODE_ROUGHNESS,
        SetRoughness => RXR_NODE_SET_ROUGHNESS, Metallic => RXR_NODE_METALLIC, SetMetallic => RXR_NODE_SET_METALLIC,
        Emissive => RXR_NODE_EMISSIVE, SetEmissive => RXR_NODE_SET_EMISSIVE, Opacity => RXR_NODE_OPACITY, SetOpacity => RXR_NODE_SET_OPACITY,
        Bump => RXR_NODE_BUMP, SetBump => RXR_NODE_SET_BUMP, Alloc => RXR_NODE_ALLOC, Iterate => RXR_NODE_ITERATE, Save => RXR_NODE_SAVE,
        PaletteIndex => RXR_NODE_PALETTE_INDEX,
        // the variants with payloads are handled by serialise_ops
        LoadGlobal(_) | StoreGlobal(_) | LoadLocal(_) | StoreLocal(_) | GetComponents(_) | SetComponents(_) | If(..) | For(..) | Push(_)
        | FunctionCall(..) | Sample | SampleNormal => unreachable!(),
    }
}

/// one program's functions as word streams (owned: the rxr_function views point into them)
struct FlatProgram {
    functions: Vec<Vec<u32>>,
    views: Vec<rxr_function>,
}

/// `rxr_set_shaders`: scene.shaders first, then every chunk's shaders in the frame's chunk order (rxr_chunk.program_base).
/// Returns Ok(program_base per chunk) or the library's status.
fn upload_programs(ctx: *mut rxr_ctx, scene: &Scene, chunk_keys: &[(i32, i32)], assets: &Assets, cache: &mut u64) -> Result<Vec<u32>, i32> {
    let mut all: Vec<&Program> = scene.shaders.iter().collect();
    let mut bases = vec![];
    for k in chunk_keys {
        bases.push(all.len() as u32);
        all.extend(scene.chunks[k].shaders.iter());
    }
    let mut uses_patterns = false;
    let mut flat: Vec<FlatProgram> = all
        .iter()
        .map(|p| {
            let functions: Vec<Vec<u32>> = p
                .user_functions
                .iter()
                .map(|f| {
                    let mut w = vec![];
                    serialise_ops(f, &mut w, &mut uses_patterns);
                    w
                })
                .collect();
            FlatProgram { functions, views: vec![] }
        })
        .collect();
    // fingerprint: the word streams + the palette (patterns are process-global and immutable once computed)
    let mut h = 1469598103934665603u64;
    for (p, f) in all.iter().zip(&flat) {
        fnv_usize(&mut h, p.globals);
        fnv_usize(&mut h, p.shade_index.map(|i| i + 1).unwrap_or(0));
        fnv_usize(&mut h, p.shade_locals);
        for w in &f.functions {
            fnv_usize(&mut h, w.len());
            for x in w {
                fnv(&mut h, &x.to_le_bytes());
            }
        }
    }
    let palette: Vec<f32> = assets.palette.colors.iter().flat_map(|c| c.as_ref().map(|c| c.to_vec3().into_array()).unwrap_or([0.0; 3])).collect();
    let present: Vec<u8> = assets.palette.colors.iter().map(|c| c.is_some() as u8).collect();
    for x in &palette {
        fnv(&mut h, &x.to_bits().to_le_bytes());
    }
    fnv(&mut h, &present);
    fnv_usize(&mut h, uses_patterns as usize);
    if h == *cache {
        return Ok(bases);
    }
    for f in flat.iter_mut() {
        f.views = f.functions.iter().map(|w| rxr_function { words: w.as_ptr(), n_words: w.len() as u32 }).collect();
    }
    let programs: Vec<rxr_program> = all
        .iter()
        .zip(&flat)
        .map(|(p, f)| rxr_program {
            n_globals: p.globals as u32,
            shade_index: p.shade_index.map(|i| i as i32).unwrap_or(-1),
            shade_locals: p.shade_locals as u32,
            functions: f.views.as_ptr(),
            n_functions: f.views.len() as u32,
        })
        .collect();
    // rusteria's process-global pattern banks (rusteria/src/textures/patterns.rs:89, :119; TexStorage.data: Vec<Vec3<f32>>,
    // vek's default Vec3 is repr(C)): only touched when a program samples them -- the first call builds all of them
    let pat = |src: &'static [rusteria::textures::TexStorage]| -> Vec<rxr_pattern> {
        src.iter().map(|t| rxr_pattern { rgb: t.data.as_ptr() as *const f32, width: t.width as u32, height: t.height as u32 }).collect()
    };
    let (patterns, normal_patterns) = if uses_patterns {
        (pat(rusteria::textures::patterns::patterns()), pat(rusteria::textures::patterns::patterns_normal()))
    } else {
        (vec![], vec![])
    };
    let set = rxr_shader_set {
        programs: programs.as_ptr(),
        n_programs: programs.len() as u32,
        patterns: patterns.as_ptr(),
        n_patterns: patterns.len() as u32,
        normal_patterns: normal_patterns.as_ptr(),
        n_normal_patterns: normal_patterns.len() as u32,
        palette_rgb: palette.as_ptr(),
        palette_present: present.as_ptr(),
        n_palette: present.len() as u32,
    };
    let rc = unsafe { rxr_set_shaders(ctx, &set) };
    if rc != RXR_OK {
        *cache = 0;
        return Err(rc);
    }
    *cache = h;
    Ok(bases)
}

// ---- flattening ------------------------------------------------------------------------------------------------
fn mat4_cols(m: &Mat4<f32>) -> [f32; 16] {
    m.into_col_array() // vek stores column-major: cols[c][r] -> m[c*4 + r]
}

/// PixelSource -> rxr_source.  EntityTile / ItemTile (src/map/pixelsource.rs:29-30) are looked up HERE, once per batch
/// (the raster loops do it per fragment: src/rasterizer.rs:1140-1187, :705-748, :1548-1595): a hit is appended to the dynamic
/// tiles (`extra`), a miss becomes RXR_SOURCE_MISSING ([0, 0, 0, 0]).
fn source_of(src: &PixelSource, assets: &Assets, dynamic_base: usize, extra: &mut Vec<*const Tile>) -> rxr_source {
    let plain = |kind: u32, index: u32, pixel: [u8; 4]| rxr_source { kind, index, pixel };
    let mut hit = |tile: Option<&Tile>| match tile {
        Some(t) => {
            let p = t as *const Tile;
            let slot = extra.iter().position(|q| *q == p).unwrap_or_else(|| {
                extra.push(p);
                extra.len() - 1
            });
            plain(RXR_SOURCE_DYNAMIC_TILE, (dynamic_base + slot) as u32, [0; 4])
        }
        None => plain(RXR_SOURCE_MISSING, 0, [0; 4]),
    };
    match src {
        PixelSource::StaticTileIndex(i) => plain(RXR_SOURCE_STATIC_TILE, *i as u32, [0; 4]),
        PixelSource::DynamicTileIndex(i) => plain(RXR_SOURCE_DYNAMIC_TILE, *i as u32, [0; 4]),
        PixelSource::Pixel(p) => plain(RXR_SOURCE_PIXEL, 0, *p),
        PixelSource::Terrain => plain(RXR_SOURCE_TERRAIN, 0, [0; 4]),
        PixelSource::EntityTile(id, index) => hit(assets.entity_tiles.get(id).and_then(|s| s.get_index(*index as usize)).map(|kv| kv.1)),
        PixelSource::ItemTile(id, index) => hit(assets.item_tiles.get(id).and_then(|s| s.get_index(*index as usize)).map(|kv| kv.1)),
        _ => plain(RXR_SOURCE_OTHER, 0, [0; 4]),
    }
}

fn light_of(l: &CompiledLight) -> rxr_light {
    rxr_light {
        light_type: l.light_type as u32, // enum order == RXR_LIGHT_* (src/map/light.rs:6-14)
        position: l.position.into_array(),
        color: l.color,
        intensity: l.intensity,
        emitting: l.emitting as u32,
        start_distance: l.start_distance,
        end_distance: l.end_distance,
        flicker: l.flicker,
        direction: l.direction.into_array(),
        cone_angle: l.cone_angle,
        normal: l.normal.into_array(),
        width: l.width,
        height: l.height,
        from_linedef: l.from_linedef as u32,
    }
}

fn occluder_of(e: &(BBox, f32)) -> rxr_occluder {
    rxr_occluder { min: e.0.min.into_array(), max: e.0.max.into_array(), occlusion: e.1 }
}

fn texture_of(t: &Texture) -> rxr_texture {
    rxr_texture { rgba: t.data.as_ptr(), width: t.width as u32, height: t.height as u32 }
}

/// owned repacks of what cannot be passed by pointer: `usize` index triples, the `visible` flags of the `Edges`, `Vec3` normals
#[derive(Default)]
struct Repack {
    indices: Vec<u32>,
    /// 3D batches cross the ABI WITHOUT their Edges records (ABI 5: `edges` NULL): one word per triangle says `visible` (a public field),
    /// the device rebuilds a / b / c from the projected vertices under `cull_mode` -- 4 instead of 40 bytes per triangle over PCIe, and
    /// nothing private of `Edges` is read (rounds 1-3 needed an accessor patched into the crate; 2D batches: the library builds
    /// `Edges::new([v0,v1,v2],[v1,v2,v0], true)` itself when `rxr_batch2d.edges` is NULL)
    edge_visible: Vec<u32>,
    normals: Vec<f32>,
}

fn cull_of(c: &CullMode) -> u32 {
    match c {
        CullMode::Off => RXR_CULL_OFF,
        CullMode::Front => RXR_CULL_FRONT,
        CullMode::Back => RXR_CULL_BACK,
    }
}

struct Item3D<'a> {
    batch: &'a Batch3D,
    list: u32,
    chunk: i32,
}
struct Item2D<'a> {
    batch: &'a Batch2D,
    chunk: i32,
}

/// `Scene::project` (src/scene.rs:154-200) for large scenes, with every 3D batch handed to the device as soon as it is projected
/// (include/rxr.h: rxr_stream_begin / rxr_stream_batch3d): the same rayon fan-out over the same batches, one job per batch, and at
/// the end of a job the batch's arrays -- projected_vertices and clipped_uvs in place, indices / Edges / normals repacked into buffers
/// that live until the frame has been uploaded -- go to `rxr_stream_batch3d`.  The library retires the batches in submission order,
/// copies them into pinned memory and ships them group by group while rayon is still projecting the rest; `rxr_upload_frame` then
/// only adds what surrounds them.  Anything the library refuses makes that later call take the frame from scratch: the result is
/// the same either way.  (To let the device PULL the arrays instead of the library copying them, register the Vecs once with
/// `rxr_pin_host_buffer` and call `rxr_stream_begin_pinned`: the 1 M-triangle grid's call then takes 6.4 ms instead of 8.7.)
fn project_streaming(this: &Rasterizer, scene: &mut Scene, ctx: *mut rxr_ctx, st: &mut Caches) {
    use rayon::prelude::*;
    // the 2D half as Scene::project does it
    for chunk in scene.chunks.values_mut() {
        chunk.batches2d.par_iter_mut().for_each(|b| b.project(this.projection_matrix_2d));
        if let Some(t) = &mut chunk.terrain_batch2d {
            t.project(this.projection_matrix_2d);
        }
    }
    scene.d2_static.par_iter_mut().for_each(|b| b.project(this.projection_matrix_2d));
    scene.d2_dynamic.par_iter_mut().for_each(|b| b.project(this.projection_matrix_2d));
    // the 3D batches in submission order (src/rasterizer.rs:314-405): the index the device knows them by
    let mut order: Vec<&mut Batch3D> = vec![];
    for chunk in scene.chunks.values_mut() {
        order.extend(chunk.batches3d_opacity.iter_mut());
        order.extend(chunk.batches3d.iter_mut());
        order.extend(chunk.terrain_batch3d.iter_mut());
    }
    order.extend(scene.d3_static.iter_mut());
    order.extend(scene.d3_dynamic.iter_mut());
    order.extend(scene.d3_overlay.iter_mut());
    // upper bounds of what the near-plane clip can produce (batch3d.rs:627-686)
    let cap_v: Vec<u32> = order.iter().map(|b| (b.vertices.len() + 4 * b.indices.len()) as u32).collect();
    let cap_t: Vec<u32> = order.iter().map(|b| (3 * b.indices.len()) as u32).collect();
    st.stream.resize_with(order.len(), Repack::default);
    st.streamed = unsafe { rxr_stream_begin(ctx, order.len() as u32, cap_v.as_ptr(), cap_t.as_ptr()) } == RXR_OK;
    let ctx_addr = ctx as usize; // (raw pointers are not Send; the calls are thread-safe)
    let streamed = st.streamed;
    let (view, proj, w, h) = (this.view_matrix, this.projection_matrix, this.width, this.height);
    order.into_par_iter().zip(st.stream.par_iter_mut()).enumerate().for_each(|(i, (b, r))| {
        b.clip_and_project(view, proj, w, h);
        r.indices.clear();
        r.indices.extend(b.clipped_indices.iter().flat_map(|&(a, bb, c)| [a as u32, bb as u32, c as u32])); // usize -> u32
        r.edge_visible.clear();
        r.edge_visible.extend(b.edges.iter().map(|e| e.visible as u32));
        r.normals.clear();
        r.normals.extend(b.clipped_normals.iter().flat_map(|n| [n.x, n.y, n.z]));
        if streamed {
            let v = rxr_batch3d {
                projected_vertices: b.projected_vertices.as_ptr() as *const f32,
                clipped_uvs: b.clipped_uvs.as_ptr() as *const f32,
                clipped_normals: if b.normals.is_empty() { std::ptr::null() } else { r.normals.as_ptr() },
                clipped_indices: r.indices.as_ptr(),
                edges: std::ptr::null(),
                edge_visible: r.edge_visible.as_ptr(),
                cull_mode: cull_of(&b.cull_mode),
                n_vertices: b.projected_vertices.len() as u32,
                n_triangles: b.edges.len() as u32,
                // (only the arrays and the counts are read at hand-over; the header travels with rxr_upload_frame)
                has_bounding_box: 0,
                bounding_box: [0.0; 4],
                repeat_mode: 0,
                source: rxr_source { kind: RXR_SOURCE_OTHER, index: 0, pixel: [0; 4] },
                ambient_color: [0.0; 3],
                shader: -1,
                has_profile_id: 0,
                profile_id: 0,
                list: RXR_LIST_STATIC,
                chunk: -1,
            };
            unsafe { rxr_stream_batch3d(ctx_addr as *mut rxr_ctx, i as u32, &v) }; // a refusal: rxr_upload_frame starts over
        }
    });
}

pub trait RasterizeHip {
    fn rasterize_hip(&mut self, scene: &mut Scene, pixels: &mut [u8], width: usize, height: usize, tile_size: usize, assets: &Assets);
}

impl RasterizeHip for Rasterizer {
    fn rasterize_hip(&mut self, scene: &mut Scene, pixels: &mut [u8], width: usize, height: usize, tile_size: usize, assets: &Assets) {
        let mut guard = STATE.lock().unwrap();
        let st = guard.get_or_insert_with(Caches::default);
        if !st.tried {
            st.tried = true;
            st.ctx = create_context();
        }
        let ctx = match &st.ctx {
            Some(c) => c.0,
            None => return self.rasterize(scene, pixels, width, height, tile_size, assets), // no GPU: the reference's CPU path
        };
        // the editor's procedural sky lives in render-graph nodes (render_setup / render_ambient_color / render_miss_d3,
        // src/rasterizer.rs:227-253, :424-432): not on the device
        if !self.render_graph.collect_nodes_from(0, 0).is_empty() || !self.render_graph.collect_nodes_from(0, 1).is_empty() {
            return self.rasterize(scene, pixels, width, height, tile_size, assets);
        }
        assert!(pixels.len() >= width * height * 4);
        let device_projection = std::env::var("RXR_DEVICE_PROJECTION").map(|v| v == "1").unwrap_or(false);

        // ---- the host half of Rasterizer::rasterize, verbatim (src/rasterizer.rs:194-223) ----
        self.width = width as f32;
        self.height = height as f32;
        self.hash_anim = rusterix::hash_u32(scene.animation_frame as u32); // hoisted and made `pub` by the patch (:199-208)
        if device_projection {
            // both halves of Scene::project run on the device: the 3D batches as rxr_set_meshes, the 2D batches as rxr_set_meshes2d
            // with this frame's Mat3 (rxr_set_projection2d); nothing is projected here
        } else {
            // a million elements and more: project and hand over batch by batch (RXR_STREAM_UPLOAD=0: the plain sequence)
            let elements: usize = scene.chunks.values().flat_map(|c| c.batches3d.iter().chain(&c.batches3d_opacity)).chain(&scene.d3_static).chain(&scene.d3_dynamic)
                .map(|b| b.vertices.len() + b.indices.len()).sum();
            st.streamed = false;
            if elements >= (1 << 20) && unsafe { rxr_member_count(ctx) } == 1 && std::env::var("RXR_STREAM_UPLOAD").map(|v| v != "0").unwrap_or(true) {
                project_streaming(self, scene, ctx, st);
            } else {
                scene.project(self.projection_matrix_2d, self.view_matrix, self.projection_matrix, self.width, self.height); // :210
            }
        }
        let mut appended = 0usize;
        for chunk in scene.chunks.values() {
            for light in &chunk.lights {
                scene.dynamic_lights.push(light.clone()); // :219-223
                appended += 1;
            }
        }

        if !device_frame(self, scene, pixels, width, height, tile_size, assets, ctx, st, device_projection) {
            // RXR_ERR_UNSUPPORTED (state leaks of the reference's per-tile Execution, a fourth nested opacity batch, an impure
            // program, ...) or a device error: this frame is rendered by the reference's own CPU loops.  rasterize() projects
            // the scene itself and appends the chunk lights again (:219-223): take ours off first, so that the CPU frame sees
            // exactly what a plain rasterize() call would.
            let keep = scene.dynamic_lights.len().saturating_sub(appended);
            scene.dynamic_lights.truncate(keep);
            self.rasterize(scene, pixels, width, height, tile_size, assets);
        }
    }
}

/// the host-projected 3D batches as the ABI sees them: arrays by pointer (the repacked ones out of `repacks`), headers from `header`
#[allow(clippy::type_complexity)]
fn batch3d_views(items3: &[Item3D], repacks: &[Repack], src3: &[rxr_source],
                 header: &dyn Fn(&Item3D, rxr_source) -> (u32, rxr_source, [f32; 3], i32, u32, u32, u32, i32)) -> Vec<rxr_batch3d> {
    items3
        .iter()
        .zip(repacks)
        .zip(src3)
        .map(|((i, r), s)| {
            let b = i.batch;
            let bb = b.bounding_box.unwrap_or(Rect { x: 0.0, y: 0.0, width: 0.0, height: 0.0 });
            let (repeat_mode, source, ambient_color, shader, has_profile_id, profile_id, list, chunk) = header(i, *s);
            rxr_batch3d {
                projected_vertices: b.projected_vertices.as_ptr() as *const f32,
                clipped_uvs: b.clipped_uvs.as_ptr() as *const f32,
                clipped_normals: if b.normals.is_empty() { std::ptr::null() } else { r.normals.as_ptr() }, // :1083
                clipped_indices: r.indices.as_ptr(),
                edges: std::ptr::null(),
                edge_visible: r.edge_visible.as_ptr(),
                cull_mode: cull_of(&b.cull_mode),
                n_vertices: b.projected_vertices.len() as u32,
                n_triangles: b.edges.len() as u32,
                has_bounding_box: b.bounding_box.is_some() as u32,
                bounding_box: [bb.x, bb.y, bb.width, bb.height],
                repeat_mode, source, ambient_color, shader, has_profile_id, profile_id, list, chunk,
            }
        })
        .collect()
}

/// Everything after the host half: flatten the projected scene, keep the device's resident data current, render.  Takes the
/// scene immutably; `false` = the caller renders this frame on the CPU.
#[allow(clippy::too_many_arguments)]
fn device_frame(this: &Rasterizer, scene: &Scene, pixels: &mut [u8], width: usize, height: usize, tile_size: usize, assets: &Assets, ctx: *mut rxr_ctx,
                st: &mut Caches, device_projection: bool) -> bool {
    {
        // ---- submission order of src/rasterizer.rs:314-405 / :503-552 (chunks in the map's iteration order) ----
        let chunk_keys: Vec<(i32, i32)> = scene.chunks.keys().copied().collect();
        let mut items3: Vec<Item3D> = vec![];
        let mut items2: Vec<Item2D> = vec![];
        for (ci, k) in chunk_keys.iter().enumerate() {
            let chunk = &scene.chunks[k];
            let ci = ci as i32;
            items3.extend(chunk.batches3d_opacity.iter().map(|b| Item3D { batch: b, list: RXR_LIST_CHUNK_OPACITY, chunk: ci }));
            items3.extend(chunk.batches3d.iter().map(|b| Item3D { batch: b, list: RXR_LIST_CHUNK, chunk: ci }));
            items3.extend(chunk.terrain_batch3d.iter().map(|b| Item3D { batch: b, list: RXR_LIST_CHUNK_TERRAIN, chunk: ci })); // :343-356
            items2.extend(chunk.batches2d.iter().map(|b| Item2D { batch: b, chunk: ci }));
            items2.extend(chunk.terrain_batch2d.iter().map(|b| Item2D { batch: b, chunk: ci })); // :515-525
        }
        items3.extend(scene.d3_static.iter().map(|b| Item3D { batch: b, list: RXR_LIST_STATIC, chunk: -1 }));
        items3.extend(scene.d3_dynamic.iter().map(|b| Item3D { batch: b, list: RXR_LIST_DYNAMIC, chunk: -1 }));
        items3.extend(scene.d3_overlay.iter().map(|b| Item3D { batch: b, list: RXR_LIST_OVERLAY, chunk: -1 }));
        items2.extend(scene.d2_static.iter().map(|b| Item2D { batch: b, chunk: -1 }));
        items2.extend(scene.d2_dynamic.iter().map(|b| Item2D { batch: b, chunk: -1 }));

        // ---- Rusteria programs (scene.shaders + chunk.shaders) ----
        let program_bases = match upload_programs(ctx, scene, &chunk_keys, assets, &mut st.shaders) {
            Ok(b) => b,
            Err(_) => return false, // e.g. an impure program
        };

        // ---- sources first: entity / item sequence tiles extend the dynamic tile table ----
        let mut extra_tiles: Vec<*const Tile> = vec![];
        let dynamic_base = scene.dynamic_textures.len();
        let src3: Vec<rxr_source> = items3.iter().map(|i| source_of(&i.batch.source, assets, dynamic_base, &mut extra_tiles)).collect();
        let src2: Vec<rxr_source> = items2.iter().map(|i| source_of(&i.batch.source, assets, dynamic_base, &mut extra_tiles)).collect();

        // ---- textures: assets.tile_list (static) + scene.dynamic_textures + resolved sequence tiles ----
        {
            let mut h = 1469598103934665603u64;
            // the fingerprint covers CONTENTS, not only addresses: static tiles by sampled probes (fnv_sampled), dynamic textures
            // and the resolved entity / item tiles -- small, rewritten per frame by the engine -- by a full hash
            let mut mix_tile = |t: &Tile, full: bool| {
                fnv_usize(&mut h, t.textures.len());
                for x in &t.textures {
                    fnv_usize(&mut h, x.data.as_ptr() as usize);
                    fnv_usize(&mut h, x.width);
                    fnv_usize(&mut h, x.height);
                    if full {
                        fnv_usize(&mut h, x.data.len());
                        fnv(&mut h, &x.data);
                    } else {
                        fnv_sampled(&mut h, &x.data);
                    }
                }
            };
            assets.tile_list.iter().for_each(|t| mix_tile(t, false));
            scene.dynamic_textures.iter().for_each(|t| mix_tile(t, true));
            extra_tiles.iter().for_each(|t| mix_tile(unsafe { &**t }, true));
            if h != st.textures {
                let mut store: Vec<Vec<rxr_texture>> = vec![];
                let mut view = |t: &Tile| -> (usize, u32) {
                    store.push(t.textures.iter().map(texture_of).collect());
                    (store.len() - 1, t.textures.len() as u32)
                };
                let s_idx: Vec<(usize, u32)> = assets.tile_list.iter().map(&mut view).collect();
                let mut d_idx: Vec<(usize, u32)> = scene.dynamic_textures.iter().map(&mut view).collect();
                d_idx.extend(extra_tiles.iter().map(|t| view(unsafe { &**t })));
                let tiles = |idx: &[(usize, u32)]| -> Vec<rxr_tile> { idx.iter().map(|(i, n)| rxr_tile { textures: store[*i].as_ptr(), n_textures: *n }).collect() };
                let (st_tiles, dy_tiles) = (tiles(&s_idx), tiles(&d_idx));
                if unsafe { rxr_set_textures(ctx, st_tiles.as_ptr(), st_tiles.len() as u32, dy_tiles.as_ptr(), dy_tiles.len() as u32) } != RXR_OK {
                    st.textures = 0;
                    return false;
                }
                st.textures = h;
            }
        }

        // ---- 3D batches: host-projected arrays, or (RXR_DEVICE_PROJECTION=1) object-space meshes registered once ----
        let header = |i: &Item3D, src: rxr_source| (i.batch.repeat_mode as u32, src, i.batch.ambient_color.into_array(), i.batch.shader.map(|s| s as i32).unwrap_or(-1),
                                                     i.batch.profile_id.is_some() as u32, i.batch.profile_id.unwrap_or(0), i.list, i.chunk);
        let mut repacks: Vec<Repack> = vec![];
        let mut b3: Vec<rxr_batch3d> = vec![];
        let mut mesh_transforms: Vec<f32> = vec![];
        if device_projection {
            let mut h = 1469598103934665603u64;
            for (i, s) in items3.iter().zip(&src3) {
                let b = i.batch;
                for v in [b.vertices.as_ptr() as usize, b.vertices.len(), b.indices.as_ptr() as usize, b.indices.len(), b.normals.len(), b.cull_mode as usize,
                          b.repeat_mode as usize, s.kind as usize, s.index as usize, i.list as usize, (i.chunk + 1) as usize,
                          b.shader.map(|x| x + 1).unwrap_or(0), b.profile_id.map(|x| x as usize + 1).unwrap_or(0)] {
                    fnv_usize(&mut h, v);
                }
                fnv(&mut h, &s.pixel);
                // ... and what the arrays hold (sampled: a registered mesh edited in place must be registered again)
                fnv_sampled(&mut h, as_bytes(&b.vertices));
                fnv_sampled(&mut h, as_bytes(&b.indices));
                fnv_sampled(&mut h, as_bytes(&b.normals));
                fnv_sampled(&mut h, as_bytes(&b.uvs));
                mesh_transforms.extend_from_slice(&mat4_cols(&b.transform_3d)); // per frame: moving objects need no re-registration
            }
            if h != st.meshes {
                for i in &items3 {
                    let b = i.batch;
                    if !b.indices.is_empty() && b.normals.len() < b.vertices.len() {
                        return false; // clip_and_project indexes this.normals (batch3d.rs:605): let the reference panic as it would
                    }
                    repacks.push(Repack {
                        indices: b.indices.iter().flat_map(|&(a, bb, c)| [a as u32, bb as u32, c as u32]).collect(),
                        edge_visible: vec![],
                        normals: b.normals.iter().flat_map(|n| [n.x, n.y, n.z]).collect(),
                    });
                }
                let meshes: Vec<rxr_mesh3d> = items3
                    .iter()
                    .zip(&repacks)
                    .zip(&src3)
                    .map(|((i, r), s)| {
                        let b = i.batch;
                        let (repeat_mode, source, ambient_color, shader, has_profile_id, profile_id, list, chunk) = header(i, *s);
                        rxr_mesh3d {
                            vertices: b.vertices.as_ptr() as *const f32,
                            indices: r.indices.as_ptr(),
                            uvs: b.uvs.as_ptr() as *const f32,
                            normals: if r.normals.is_empty() { std::ptr::null() } else { r.normals.as_ptr() },
                            n_vertices: b.vertices.len() as u32,
                            n_triangles: b.indices.len() as u32,
                            transform_3d: mat4_cols(&b.transform_3d),
                            cull_mode: b.cull_mode as u32,
                            repeat_mode, source, ambient_color, shader, has_profile_id, profile_id, list, chunk,
                        }
                    })
                    .collect();
                if unsafe { rxr_set_meshes(ctx, meshes.as_ptr(), meshes.len() as u32) } != RXR_OK {
                    st.meshes = 0;
                    return false;
                }
                st.meshes = h;
            }
        } else if st.streamed && st.stream.len() == items3.len() {
            // project_streaming repacked these batches and gave the device their addresses: the frame names the same buffers
            repacks = std::mem::take(&mut st.stream);
            b3 = batch3d_views(&items3, &repacks, &src3, &header);
        } else {
            repacks = items3
                .iter()
                .map(|i| Repack {
                    indices: i.batch.clipped_indices.iter().flat_map(|&(a, bb, c)| [a as u32, bb as u32, c as u32]).collect(), // usize -> u32
                    edge_visible: i.batch.edges.iter().map(|e| e.visible as u32).collect(),
                    normals: i.batch.clipped_normals.iter().flat_map(|n| [n.x, n.y, n.z]).collect(),
                })
                .collect();
            b3 = batch3d_views(&items3, &repacks, &src3, &header);
        }

        // ---- 2D batches on the device (RXR_DEVICE_PROJECTION=1): object-space arrays registered when they change, the Mat3 per frame ----
        let idx2: Vec<Vec<u32>> = if device_projection {
            items2.iter().map(|i| i.batch.indices.iter().flat_map(|&(a, bb, c)| [a as u32, bb as u32, c as u32]).collect()).collect()
        } else {
            vec![]
        };
        if device_projection {
            let mut h = 1469598103934665603u64;
            for (i, s) in items2.iter().zip(&src2) {
                let b = i.batch;
                for v in [b.vertices.len(), b.indices.len(), b.mode as usize, b.repeat_mode as usize, s.kind as usize, s.index as usize, b.receives_light as usize,
                          b.shader.map(|x| x + 1).unwrap_or(0), (i.chunk + 1) as usize] {
                    fnv_usize(&mut h, v);
                }
                fnv(&mut h, &s.pixel);
                fnv(&mut h, as_bytes(&b.vertices)); // 2D batches are small: the whole contents
                fnv(&mut h, as_bytes(&b.uvs));
                fnv(&mut h, as_bytes(&b.indices));
            }
            if h != st.meshes2d {
                let m2: Vec<rxr_mesh2d> = items2
                    .iter()
                    .zip(&idx2)
                    .zip(&src2)
                    .map(|((i, ix), s)| {
                        let b = i.batch;
                        rxr_mesh2d {
                            vertices: b.vertices.as_ptr() as *const f32,
                            indices: ix.as_ptr(),
                            uvs: b.uvs.as_ptr() as *const f32,
                            n_vertices: b.vertices.len() as u32,
                            n_triangles: b.indices.len() as u32,
                            mode: b.mode as u32,
                            repeat_mode: b.repeat_mode as u32,
                            source: *s,
                            receives_light: b.receives_light as u32,
                            shader: b.shader.map(|x| x as i32).unwrap_or(-1),
                            chunk: i.chunk,
                        }
                    })
                    .collect();
                if unsafe { rxr_set_meshes2d(ctx, m2.as_ptr(), m2.len() as u32) } != RXR_OK {
                    st.meshes2d = 0;
                    return false;
                }
                st.meshes2d = h;
            }
            // Option<Mat3<f32>> in vek's column-major order (cols[c][r])
            let m3: Option<[f32; 9]> = this.projection_matrix_2d.map(|m| {
                let c = m.into_col_array();
                [c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]]
            });
            unsafe { rxr_set_projection2d(ctx, m3.as_ref().map(|a| a.as_ptr()).unwrap_or(std::ptr::null())) };
        }

        // ---- 2D batches (host-projected; none when the device projects them) ----
        let items2: Vec<Item2D> = if device_projection { vec![] } else { items2 };
        let repacks2: Vec<Repack> = items2
            .iter()
            .map(|i| Repack {
                indices: i.batch.indices.iter().flat_map(|&(a, bb, c)| [a as u32, bb as u32, c as u32]).collect(),
                edge_visible: vec![],
                normals: vec![],
            })
            .collect();
        let b2: Vec<rxr_batch2d> = items2
            .iter()
            .zip(&repacks2)
            .zip(&src2)
            .map(|((i, r), s)| {
                let b = i.batch;
                let bb = b.bounding_box.unwrap_or(Rect { x: 0.0, y: 0.0, width: 0.0, height: 0.0 });
                rxr_batch2d {
                    projected_vertices: b.projected_vertices.as_ptr() as *const f32,
                    uvs: b.uvs.as_ptr() as *const f32,
                    indices: r.indices.as_ptr(),
                    edges: std::ptr::null(), // (ABI 5: Edges::new of the projected vertices is built by the library)
                    n_vertices: b.projected_vertices.len() as u32,
                    n_triangles: b.indices.len() as u32,
                    has_bounding_box: b.bounding_box.is_some() as u32,
                    bounding_box: [bb.x, bb.y, bb.width, bb.height],
                    mode: b.mode as u32,
                    repeat_mode: b.repeat_mode as u32,
                    source: *s,
                    receives_light: b.receives_light as u32,
                    shader: b.shader.map(|x| x as i32).unwrap_or(-1),
                    chunk: i.chunk,
                }
            })
            .collect();

        // ---- lights, occluders, linedefs, chunks ----
        let lights: Vec<rxr_light> = scene.lights.iter().chain(&scene.dynamic_lights).map(light_of).collect(); // :1373
        // MapMini.occluded_sectors / .linedefs are private: accessors from the patch in INTEGRATION.md
        let occluders: Vec<rxr_occluder> = this.mapmini.occluded_sectors().iter().map(occluder_of).collect();
        let linedefs: Vec<rxr_linedef> = this.mapmini.linedefs().iter().map(|l| rxr_linedef { start: l.start.into_array(), end: l.end.into_array() }).collect();
        let chunk_occ: Vec<Vec<rxr_occluder>> = chunk_keys.iter().map(|k| scene.chunks[k].occluded_sectors.iter().map(occluder_of).collect()).collect();
        // chunk.shader_textures: Vec<Option<Texture>> (src/chunk.rs:53); rgba == NULL for None
        let chunk_baked: Vec<Vec<rxr_texture>> = chunk_keys
            .iter()
            .map(|k| scene.chunks[k].shader_textures.iter().map(|t| t.as_ref().map(texture_of).unwrap_or(rxr_texture { rgba: std::ptr::null(), width: 0, height: 0 })).collect())
            .collect();
        let chunk_terrain: Vec<Option<rxr_texture>> = chunk_keys.iter().map(|k| scene.chunks[k].terrain_texture.as_ref().map(texture_of)).collect();
        let chunks: Vec<rxr_chunk> = chunk_keys
            .iter()
            .enumerate()
            .map(|(ci, k)| {
                let c = &scene.chunks[k];
                rxr_chunk {
                    occluders: chunk_occ[ci].as_ptr(),
                    n_occluders: chunk_occ[ci].len() as u32,
                    program_base: program_bases[ci],
                    n_programs: c.shaders.len() as u32,
                    shader_textures: chunk_baked[ci].as_ptr(),
                    n_shader_textures: chunk_baked[ci].len() as u32,
                    terrain_texture: chunk_terrain[ci].as_ref().map(|t| t as *const rxr_texture).unwrap_or(std::ptr::null()),
                    origin: c.origin.into_array(),
                    size: c.size,
                }
            })
            .collect();

        // background: any `dyn Shader` is evaluated here by the reference's own code and handed over as pixels (bit-identical by
        // construction).  The device can evaluate VGrayGradientShader (RXR_BG_VGRADIENT) and GridShader (RXR_BG_GRID +
        // background_grid) itself, but a `Box<dyn Shader>` does not tell which one it is: that needs a `kind()` / parameter
        // accessor on the trait (a two-line patch to src/shader/mod.rs), after which the pixel loop below is skipped for them.
        // In 3D mode the background is overwritten by the resolve loop anyway (src/rasterizer.rs:420-461).
        let mut bg_pixels: Vec<u8> = vec![];
        let mut background_kind = RXR_BG_NONE;
        if !this.render_mode.ignore_background_shader && !this.render_mode.supports3d() {
            if let Some(shader) = &scene.background {
                background_kind = RXR_BG_HOST_PIXELS;
                bg_pixels.reserve(width * height * 4);
                let screen = vek::Vec2::new(width as f32, height as f32);
                for y in 0..height {
                    for x in 0..width {
                        bg_pixels.extend_from_slice(&shader.shade_pixel(vek::Vec2::new(x as f32 / screen.x, y as f32 / screen.y), screen)); // :292-308
                    }
                }
            }
        }

        let (translationd2, scaled2) = this.d2_transform(); // accessor added by the patch (private fields, :68-69)
        let frame = rxr_frame {
            abi_version: RXR_ABI_VERSION,
            width: width as u32,
            height: height as u32,
            tile_size: tile_size as u32,
            inverse_view: mat4_cols(&this.inverse_view_matrix),
            inverse_projection: mat4_cols(&this.inverse_projection_matrix),
            camera_pos: this.camera_pos.into_array(),
            translationd2: translationd2.into_array(),
            scaled2,
            hash_anim: this.hash_anim,
            animation_frame: scene.animation_frame as u64,
            flags: (this.render_mode.supports2d() as u32 * RXR_FLAG_D2_ACTIVE)
                | (this.render_mode.supports3d() as u32 * RXR_FLAG_D3_ACTIVE)
                | (this.render_mode.ignore_background_shader as u32 * RXR_FLAG_IGNORE_BG_SHADER)
                | (this.preserve_transparency as u32 * RXR_FLAG_PRESERVE_TRANSPARENCY)
                | (this.background_color.is_some() as u32 * RXR_FLAG_HAS_BACKGROUND_COLOR)
                | (this.ambient_color.is_some() as u32 * RXR_FLAG_HAS_AMBIENT)
                | (this.sun_dir.is_some() as u32 * RXR_FLAG_HAS_SUN),
            background_color: this.background_color.unwrap_or([0; 4]),
            ambient: this.ambient_color.map(|a| a.into_array()).unwrap_or([0.0; 4]),
            sun_dir: this.sun_dir.map(|s| s.into_array()).unwrap_or([0.0; 3]),
            day_factor: this.day_factor,
            sample_mode: this.sample_mode as u32,
            time: this.time,
            background_kind,
            background_pixels: if bg_pixels.is_empty() { std::ptr::null() } else { bg_pixels.as_ptr() },
            batches3d: b3.as_ptr(),
            n_batches3d: b3.len() as u32,
            batches2d: b2.as_ptr(),
            n_batches2d: b2.len() as u32,
            lights: lights.as_ptr(),
            n_lights: lights.len() as u32,
            occluders: occluders.as_ptr(),
            n_occluders: occluders.len() as u32,
            linedefs: linedefs.as_ptr(),
            n_linedefs: linedefs.len() as u32,
            chunks: chunks.as_ptr(),
            n_chunks: chunks.len() as u32,
            n_shader_programs: scene.shaders.len() as u32,
            use_meshes: if device_projection { 3 } else { 0 }, // bit 0: the 3D batches, bit 1: the 2D batches are registered meshes
            view: if device_projection { mat4_cols(&this.view_matrix) } else { [0.0; 16] },
            projection: if device_projection { mat4_cols(&this.projection_matrix) } else { [0.0; 16] },
            mesh_transforms: if device_projection && !mesh_transforms.is_empty() { mesh_transforms.as_ptr() } else { std::ptr::null() },
            background_grid: [30.0, 2.0, 0.0, 0.0],
            has_brush_preview: this.brush_preview.is_some() as u32,
            brush_position: this.brush_preview.as_ref().map(|b| b.position.into_array()).unwrap_or([0.0; 3]),
            brush_radius: this.brush_preview.as_ref().map(|b| b.radius).unwrap_or(0.0),
            brush_falloff: this.brush_preview.as_ref().map(|b| b.falloff).unwrap_or(0.0),
        };
        unsafe { rxr_rasterize(ctx, &frame, pixels.as_mut_ptr()) == RXR_OK }
    }
}
