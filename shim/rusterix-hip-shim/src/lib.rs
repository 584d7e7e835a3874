//! `RasterizerHip`: drop-in for `rusterix::Rasterizer::rasterize` that keeps scene set-up,
//! `Scene::project` and the `Edges` precompute in Rust (reference src/rasterizer.rs:185-223) and hands
//! everything after it (`:256-579`) to the gfx950 kernels through include/rxr.h.
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
//! When no GPU is present (`rxr_create` fails) or the scene uses a feature the device path does not
//! implement (`RXR_ERR_UNSUPPORTED`: Rusteria shader programs), the call falls back to the
//! reference's own CPU `rasterize`, so the examples keep working everywhere.  That fallback lives
//! HERE, in the caller's crate -- the library itself never falls back.
pub mod ffi;

use ffi::*;
use rusterix::prelude::*;
use std::sync::Mutex;
use vek::Mat4;

struct Ctx(*mut rxr_ctx);
unsafe impl Send for Ctx {}
static CTX: Mutex<Option<Ctx>> = Mutex::new(None);
static TEXTURE_STAMP: Mutex<(usize, usize)> = Mutex::new((0, 0)); // (assets ptr, tile count): re-upload when it changes

fn mat4_cols(m: &Mat4<f32>) -> [f32; 16] {
    // vek stores column-major: cols[c][r] -> m[c*4 + r]
    let c = m.into_col_array();
    c
}

fn source_of(src: &PixelSource, assets: &Assets, dynamic_base: usize, extra_tiles: &mut Vec<*const Tile>) -> rxr_source {
    match src {
        PixelSource::StaticTileIndex(i) => rxr_source { kind: RXR_SOURCE_STATIC_TILE, index: *i as u32, pixel: [0; 4] },
        PixelSource::DynamicTileIndex(i) => rxr_source { kind: RXR_SOURCE_DYNAMIC_TILE, index: *i as u32, pixel: [0; 4] },
        PixelSource::Pixel(p) => rxr_source { kind: RXR_SOURCE_PIXEL, index: 0, pixel: *p },
        PixelSource::Terrain => rxr_source { kind: RXR_SOURCE_TERRAIN, index: 0, pixel: [0; 4] },
        // hash lookups are resolved on the host (src/rasterizer.rs:1140-1187): a hit is appended to
        // the dynamic tile table, a miss becomes RXR_SOURCE_MISSING ([0,0,0,0])
        PixelSource::EntityTile(id, index) => match assets.entity_tiles.get(id).and_then(|s| s.get_index(*index as usize)) {
            Some((_, tile)) => {
                extra_tiles.push(tile as *const Tile);
                rxr_source { kind: RXR_SOURCE_DYNAMIC_TILE, index: (dynamic_base + extra_tiles.len() - 1) as u32, pixel: [0; 4] }
            }
            None => rxr_source { kind: RXR_SOURCE_MISSING, index: 0, pixel: [0; 4] },
        },
        PixelSource::ItemTile(id, index) => match assets.item_tiles.get(id).and_then(|s| s.get_index(*index as usize)) {
            Some((_, tile)) => {
                extra_tiles.push(tile as *const Tile);
                rxr_source { kind: RXR_SOURCE_DYNAMIC_TILE, index: (dynamic_base + extra_tiles.len() - 1) as u32, pixel: [0; 4] }
            }
            None => rxr_source { kind: RXR_SOURCE_MISSING, index: 0, pixel: [0; 4] },
        },
        _ => rxr_source { kind: RXR_SOURCE_OTHER, index: 0, pixel: [0; 4] },
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

/// Owned, flattened copies of what cannot be passed by pointer (usize indices, private Edges).
#[derive(Default)]
struct Flat3D {
    indices: Vec<u32>,
    edges: Vec<rxr_edges>,
    normals: Vec<f32>,
}

fn flatten3d(b: &Batch3D) -> Flat3D {
    let mut f = Flat3D::default();
    f.indices.reserve(b.clipped_indices.len() * 3);
    for &(i0, i1, i2) in &b.clipped_indices {
        f.indices.extend_from_slice(&[i0 as u32, i1 as u32, i2 as u32]); // usize -> u32 (asserted < 2^32 by the ABI)
    }
    f.edges = b.edges.iter().map(|e| {
        let (a, bb, c) = e.coefficients(); // accessor added by the patch in INTEGRATION.md
        rxr_edges { a, b: bb, c, visible: e.visible as u32 }
    }).collect();
    f.normals = b.clipped_normals.iter().flat_map(|n| [n.x, n.y, n.z]).collect();
    f
}

pub trait RasterizeHip {
    fn rasterize_hip(&mut self, scene: &mut Scene, pixels: &mut [u8], width: usize, height: usize, tile_size: usize, assets: &Assets);
}

impl RasterizeHip for Rasterizer {
    fn rasterize_hip(&mut self, scene: &mut Scene, pixels: &mut [u8], width: usize, height: usize, tile_size: usize, assets: &Assets) {
        let ctx = {
            let mut g = CTX.lock().unwrap();
            if g.is_none() {
                let mut p: *mut rxr_ctx = std::ptr::null_mut();
                let dev = std::env::var("RXR_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
                if unsafe { rxr_create(&mut p, dev) } == RXR_OK {
                    *g = Some(Ctx(p));
                }
            }
            match g.as_ref() {
                Some(c) => c.0,
                None => return self.rasterize(scene, pixels, width, height, tile_size, assets), // no GPU: reference CPU path
            }
        };
        if !scene.shaders.is_empty() {
            return self.rasterize(scene, pixels, width, height, tile_size, assets); // N2 not on the device yet
        }

        // ---- the host half of Rasterizer::rasterize, verbatim (src/rasterizer.rs:194-223) ----
        self.width = width as f32;
        self.height = height as f32;
        self.hash_anim = rusterix::hash_u32(scene.animation_frame as u32); // made `pub` by the patch
        scene.project(self.projection_matrix_2d, self.view_matrix, self.projection_matrix, self.width, self.height);
        for chunk in scene.chunks.values() {
            for light in &chunk.lights {
                scene.dynamic_lights.push(light.clone());
            }
        }

        // ---- flatten ----
        let mut extra_tiles: Vec<*const Tile> = vec![];
        let dynamic_base = scene.dynamic_textures.len();
        let mut flats: Vec<Flat3D> = vec![];
        let mut b3: Vec<rxr_batch3d> = vec![];
        let mut b2_idx: Vec<Vec<u32>> = vec![];
        let mut b2_edges: Vec<Vec<rxr_edges>> = vec![];
        let mut b2: Vec<rxr_batch2d> = vec![];
        let mut chunk_occ: Vec<Vec<rxr_occluder>> = vec![];

        let mut push3d = |b: &Batch3D, list: u32, chunk: i32, flats: &mut Vec<Flat3D>, b3: &mut Vec<rxr_batch3d>,
                          extra: &mut Vec<*const Tile>| {
            flats.push(flatten3d(b));
            let f = flats.last().unwrap();
            let bb = b.bounding_box.unwrap_or_default();
            b3.push(rxr_batch3d {
                projected_vertices: b.projected_vertices.as_ptr() as *const f32,
                clipped_uvs: b.clipped_uvs.as_ptr() as *const f32,
                clipped_normals: if b.normals.is_empty() { std::ptr::null() } else { f.normals.as_ptr() },
                clipped_indices: f.indices.as_ptr(),
                edges: f.edges.as_ptr(),
                n_vertices: b.projected_vertices.len() as u32,
                n_triangles: b.edges.len() as u32,
                has_bounding_box: b.bounding_box.is_some() as u32,
                bounding_box: [bb.x, bb.y, bb.width, bb.height],
                repeat_mode: b.repeat_mode as u32,
                source: source_of(&b.source, assets, dynamic_base, extra),
                ambient_color: b.ambient_color.into_array(),
                shader: b.shader.map(|s| s as i32).unwrap_or(-1),
                has_profile_id: b.profile_id.is_some() as u32,
                profile_id: b.profile_id.unwrap_or(0),
                list,
                chunk,
            });
        };
        // submission order of src/rasterizer.rs:314-405 (chunks in the map's iteration order)
        for (ci, chunk) in scene.chunks.values().enumerate() {
            for b in &chunk.batches3d_opacity { push3d(b, RXR_LIST_CHUNK_OPACITY, ci as i32, &mut flats, &mut b3, &mut extra_tiles); }
            for b in &chunk.batches3d { push3d(b, RXR_LIST_CHUNK, ci as i32, &mut flats, &mut b3, &mut extra_tiles); }
            chunk_occ.push(chunk.occluded_sectors.iter().map(|(bb, o)| rxr_occluder { min: bb.min.into_array(), max: bb.max.into_array(), occlusion: *o }).collect());
        }
        for b in &scene.d3_static { push3d(b, RXR_LIST_STATIC, -1, &mut flats, &mut b3, &mut extra_tiles); }
        for b in &scene.d3_dynamic { push3d(b, RXR_LIST_DYNAMIC, -1, &mut flats, &mut b3, &mut extra_tiles); }
        for b in &scene.d3_overlay { push3d(b, RXR_LIST_OVERLAY, -1, &mut flats, &mut b3, &mut extra_tiles); }

        let mut push2d = |b: &Batch2D, chunk: i32| {
            b2_idx.push(b.indices.iter().flat_map(|&(a, bb, c)| [a as u32, bb as u32, c as u32]).collect());
            b2_edges.push(b.edges.iter().map(|e| { let (a, bb, c) = e.coefficients(); rxr_edges { a, b: bb, c, visible: e.visible as u32 } }).collect());
            let bb = b.bounding_box.unwrap_or_default();
            b2.push(rxr_batch2d {
                projected_vertices: b.projected_vertices.as_ptr() as *const f32,
                uvs: b.uvs.as_ptr() as *const f32,
                indices: b2_idx.last().unwrap().as_ptr(),
                edges: b2_edges.last().unwrap().as_ptr(),
                n_vertices: b.projected_vertices.len() as u32,
                n_triangles: b.indices.len() as u32,
                has_bounding_box: b.bounding_box.is_some() as u32,
                bounding_box: [bb.x, bb.y, bb.width, bb.height],
                mode: b.mode as u32,
                repeat_mode: b.repeat_mode as u32,
                source: source_of(&b.source, assets, dynamic_base, &mut extra_tiles),
                receives_light: b.receives_light as u32,
                shader: b.shader.map(|s| s as i32).unwrap_or(-1),
                chunk,
            });
        };
        for (ci, chunk) in scene.chunks.values().enumerate() { for b in &chunk.batches2d { push2d(b, ci as i32); } }
        for b in &scene.d2_static { push2d(b, -1); }
        for b in &scene.d2_dynamic { push2d(b, -1); }

        let lights: Vec<rxr_light> = scene.lights.iter().chain(&scene.dynamic_lights).map(light_of).collect();
        let occluders: Vec<rxr_occluder> = self.mapmini.occluded_sectors.iter()
            .map(|(bb, o)| rxr_occluder { min: bb.min.into_array(), max: bb.max.into_array(), occlusion: *o }).collect();
        let linedefs: Vec<rxr_linedef> = self.mapmini.linedefs.iter()
            .map(|l| rxr_linedef { start: l.start.into_array(), end: l.end.into_array() }).collect();
        let chunks: Vec<rxr_chunk> = chunk_occ.iter().map(|v| rxr_chunk { occluders: v.as_ptr(), n_occluders: v.len() as u32 }).collect();

        // ---- textures: assets.tile_list (static) + scene.dynamic_textures + resolved entity/item tiles ----
        let tile_view = |t: &Tile, store: &mut Vec<Vec<rxr_texture>>| -> rxr_tile {
            store.push(t.textures.iter().map(|x| rxr_texture { rgba: x.data.as_ptr(), width: x.width as u32, height: x.height as u32 }).collect());
            let v = store.last().unwrap();
            rxr_tile { textures: v.as_ptr(), n_textures: v.len() as u32 }
        };
        let stamp = (assets as *const Assets as usize, assets.tile_list.len() + scene.dynamic_textures.len() + extra_tiles.len());
        if *TEXTURE_STAMP.lock().unwrap() != stamp {
            let mut store = vec![];
            let st: Vec<rxr_tile> = assets.tile_list.iter().map(|t| tile_view(t, &mut store)).collect();
            let mut dy: Vec<rxr_tile> = scene.dynamic_textures.iter().map(|t| tile_view(t, &mut store)).collect();
            for t in &extra_tiles { dy.push(tile_view(unsafe { &**t }, &mut store)); }
            if unsafe { rxr_set_textures(ctx, st.as_ptr(), st.len() as u32, dy.as_ptr(), dy.len() as u32) } != RXR_OK {
                return self.rasterize(scene, pixels, width, height, tile_size, assets);
            }
            *TEXTURE_STAMP.lock().unwrap() = stamp;
        }

        // background: any `dyn Shader` is evaluated here by the reference's own code and handed over as pixels (bit-identical by
        // construction).  The device can evaluate VGrayGradientShader (RXR_BG_VGRADIENT) and GridShader (RXR_BG_GRID +
        // background_grid) itself, but a `Box<dyn Shader>` does not tell which one it is: that needs a `kind()` / parameter
        // accessor on the trait (a two-line patch to src/shader/mod.rs), after which the pixel loop below is skipped for them.
        let mut bg_pixels: Vec<u8> = vec![];
        let mut background_kind = RXR_BG_NONE;
        if !self.render_mode.ignore_background_shader && !self.render_mode.supports3d() {
            if let Some(shader) = &scene.background {
                background_kind = RXR_BG_HOST_PIXELS;
                bg_pixels.reserve(width * height * 4);
                let screen = vek::Vec2::new(width as f32, height as f32);
                for y in 0..height { for x in 0..width {
                    bg_pixels.extend_from_slice(&shader.shade_pixel(vek::Vec2::new(x as f32 / screen.x, y as f32 / screen.y), screen));
                } }
            }
        } // in 3D mode the background is overwritten by the resolve loop (src/rasterizer.rs:420-461)

        let (translationd2, scaled2) = self.d2_transform(); // accessor added by the patch
        let frame = rxr_frame {
            abi_version: RXR_ABI_VERSION,
            width: width as u32, height: height as u32, tile_size: tile_size as u32,
            inverse_view: mat4_cols(&self.inverse_view_matrix),
            inverse_projection: mat4_cols(&self.inverse_projection_matrix),
            camera_pos: self.camera_pos.into_array(),
            translationd2: translationd2.into_array(), scaled2,
            hash_anim: self.hash_anim, animation_frame: scene.animation_frame as u64,
            flags: (self.render_mode.supports2d() as u32 * RXR_FLAG_D2_ACTIVE) | (self.render_mode.supports3d() as u32 * RXR_FLAG_D3_ACTIVE)
                | (self.render_mode.ignore_background_shader as u32 * RXR_FLAG_IGNORE_BG_SHADER)
                | (self.preserve_transparency as u32 * RXR_FLAG_PRESERVE_TRANSPARENCY)
                | (self.background_color.is_some() as u32 * RXR_FLAG_HAS_BACKGROUND_COLOR)
                | (self.ambient_color.is_some() as u32 * RXR_FLAG_HAS_AMBIENT) | (self.sun_dir.is_some() as u32 * RXR_FLAG_HAS_SUN),
            background_color: self.background_color.unwrap_or([0; 4]),
            ambient: self.ambient_color.map(|a| a.into_array()).unwrap_or([0.0; 4]),
            sun_dir: self.sun_dir.map(|s| s.into_array()).unwrap_or([0.0; 3]),
            day_factor: self.day_factor,
            sample_mode: self.sample_mode as u32, time: self.time,
            background_kind, background_pixels: if bg_pixels.is_empty() { std::ptr::null() } else { bg_pixels.as_ptr() },
            batches3d: b3.as_ptr(), n_batches3d: b3.len() as u32,
            batches2d: b2.as_ptr(), n_batches2d: b2.len() as u32,
            lights: lights.as_ptr(), n_lights: lights.len() as u32,
            occluders: occluders.as_ptr(), n_occluders: occluders.len() as u32,
            linedefs: linedefs.as_ptr(), n_linedefs: linedefs.len() as u32,
            chunks: chunks.as_ptr(), n_chunks: chunks.len() as u32,
            n_shader_programs: scene.shaders.len() as u32,
            use_meshes: 0, view: [0.0; 16], projection: [0.0; 16], mesh_transforms: std::ptr::null(),
            background_grid: [30.0, 2.0, 0.0, 0.0],
            has_brush_preview: self.brush_preview.is_some() as u32,
            brush_position: self.brush_preview.as_ref().map(|b| b.position.into_array()).unwrap_or([0.0; 3]),
            brush_radius: self.brush_preview.as_ref().map(|b| b.radius).unwrap_or(0.0),
            brush_falloff: self.brush_preview.as_ref().map(|b| b.falloff).unwrap_or(0.0),
        };
        assert!(pixels.len() >= width * height * 4);
        let rc = unsafe { rxr_rasterize(ctx, &frame, pixels.as_mut_ptr()) };
        if rc != RXR_OK {
            // RXR_ERR_UNSUPPORTED / device error: render this frame with the reference's CPU loops instead
            self.rasterize(scene, pixels, width, height, tile_size, assets);
        }
    }
}
