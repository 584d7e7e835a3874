//! Raw bindings of include/rxr.h (ABI version 3).  Field order and types mirror the C header exactly.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RXR_ABI_VERSION: u32 = 3;
pub const RXR_OK: c_int = 0;

pub const RXR_FLAG_D2_ACTIVE: u32 = 1 << 0;
pub const RXR_FLAG_D3_ACTIVE: u32 = 1 << 1;
pub const RXR_FLAG_IGNORE_BG_SHADER: u32 = 1 << 2;
pub const RXR_FLAG_PRESERVE_TRANSPARENCY: u32 = 1 << 3;
pub const RXR_FLAG_HAS_BACKGROUND_COLOR: u32 = 1 << 4;
pub const RXR_FLAG_HAS_AMBIENT: u32 = 1 << 5;
pub const RXR_FLAG_HAS_SUN: u32 = 1 << 6;

pub const RXR_SOURCE_OTHER: u32 = 0;
pub const RXR_SOURCE_STATIC_TILE: u32 = 1;
pub const RXR_SOURCE_DYNAMIC_TILE: u32 = 2;
pub const RXR_SOURCE_PIXEL: u32 = 3;
pub const RXR_SOURCE_TERRAIN: u32 = 4;
pub const RXR_SOURCE_MISSING: u32 = 5;

pub const RXR_LIST_CHUNK_OPACITY: u32 = 0;
pub const RXR_LIST_CHUNK: u32 = 1;
pub const RXR_LIST_STATIC: u32 = 3;
pub const RXR_LIST_DYNAMIC: u32 = 4;
pub const RXR_LIST_OVERLAY: u32 = 5;

pub const RXR_BG_NONE: u32 = 0;
pub const RXR_BG_VGRADIENT: u32 = 1;
pub const RXR_BG_HOST_PIXELS: u32 = 2;
pub const RXR_BG_GRID: u32 = 3;

#[repr(C)]
pub struct rxr_ctx {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rxr_texture {
    pub rgba: *const u8,
    pub width: u32,
    pub height: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rxr_tile {
    pub textures: *const rxr_texture,
    pub n_textures: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rxr_light {
    pub light_type: u32,
    pub position: [f32; 3],
    pub color: [f32; 3],
    pub intensity: f32,
    pub emitting: u32,
    pub start_distance: f32,
    pub end_distance: f32,
    pub flicker: f32,
    pub direction: [f32; 3],
    pub cone_angle: f32,
    pub normal: [f32; 3],
    pub width: f32,
    pub height: f32,
    pub from_linedef: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rxr_edges {
    pub a: [f32; 3],
    pub b: [f32; 3],
    pub c: [f32; 3],
    pub visible: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rxr_source {
    pub kind: u32,
    pub index: u32,
    pub pixel: [u8; 4],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rxr_batch3d {
    pub projected_vertices: *const f32,
    pub clipped_uvs: *const f32,
    pub clipped_normals: *const f32,
    pub clipped_indices: *const u32,
    pub edges: *const rxr_edges,
    pub n_vertices: u32,
    pub n_triangles: u32,
    pub has_bounding_box: u32,
    pub bounding_box: [f32; 4],
    pub repeat_mode: u32,
    pub source: rxr_source,
    pub ambient_color: [f32; 3],
    pub shader: i32,
    pub has_profile_id: u32,
    pub profile_id: u32,
    pub list: u32,
    pub chunk: i32,
}

/// Object-space Batch3D for device-side projection (rxr_set_meshes).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rxr_mesh3d {
    pub vertices: *const f32,
    pub indices: *const u32,
    pub uvs: *const f32,
    pub normals: *const f32,
    pub n_vertices: u32,
    pub n_triangles: u32,
    pub transform_3d: [f32; 16],
    pub cull_mode: u32,
    pub repeat_mode: u32,
    pub source: rxr_source,
    pub ambient_color: [f32; 3],
    pub shader: i32,
    pub has_profile_id: u32,
    pub profile_id: u32,
    pub list: u32,
    pub chunk: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rxr_batch2d {
    pub projected_vertices: *const f32,
    pub uvs: *const f32,
    pub indices: *const u32,
    pub edges: *const rxr_edges,
    pub n_vertices: u32,
    pub n_triangles: u32,
    pub has_bounding_box: u32,
    pub bounding_box: [f32; 4],
    pub mode: u32,
    pub repeat_mode: u32,
    pub source: rxr_source,
    pub receives_light: u32,
    pub shader: i32,
    pub chunk: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rxr_occluder {
    pub min: [f32; 2],
    pub max: [f32; 2],
    pub occlusion: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rxr_linedef {
    pub start: [f32; 2],
    pub end: [f32; 2],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rxr_chunk {
    pub occluders: *const rxr_occluder,
    pub n_occluders: u32,
    pub program_base: u32,
    pub n_programs: u32,
    pub shader_textures: *const rxr_texture,
    pub n_shader_textures: u32,
    pub terrain_texture: *const rxr_texture,
    pub origin: [i32; 2],
    pub size: i32,
}

#[repr(C)]
pub struct rxr_frame {
    pub abi_version: u32,
    pub width: u32,
    pub height: u32,
    pub tile_size: u32,
    pub inverse_view: [f32; 16],
    pub inverse_projection: [f32; 16],
    pub camera_pos: [f32; 3],
    pub translationd2: [f32; 2],
    pub scaled2: f32,
    pub hash_anim: u32,
    pub animation_frame: u64,
    pub flags: u32,
    pub background_color: [u8; 4],
    pub ambient: [f32; 4],
    pub sun_dir: [f32; 3],
    pub day_factor: f32,
    pub sample_mode: u32,
    pub time: f32,
    pub background_kind: u32,
    pub background_pixels: *const u8,
    pub batches3d: *const rxr_batch3d,
    pub n_batches3d: u32,
    pub batches2d: *const rxr_batch2d,
    pub n_batches2d: u32,
    pub lights: *const rxr_light,
    pub n_lights: u32,
    pub occluders: *const rxr_occluder,
    pub n_occluders: u32,
    pub linedefs: *const rxr_linedef,
    pub n_linedefs: u32,
    pub chunks: *const rxr_chunk,
    pub n_chunks: u32,
    pub n_shader_programs: u32,
    pub use_meshes: u32,
    pub view: [f32; 16],
    pub projection: [f32; 16],
    pub mesh_transforms: *const f32,
    pub background_grid: [f32; 4],
    pub has_brush_preview: u32,
    pub brush_position: [f32; 3],
    pub brush_radius: f32,
    pub brush_falloff: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rxr_function {
    pub words: *const u32,
    pub n_words: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rxr_program {
    pub n_globals: u32,
    pub shade_index: i32,
    pub shade_locals: u32,
    pub functions: *const rxr_function,
    pub n_functions: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rxr_pattern {
    pub rgb: *const f32,
    pub width: u32,
    pub height: u32,
}

#[repr(C)]
pub struct rxr_shader_set {
    pub programs: *const rxr_program,
    pub n_programs: u32,
    pub patterns: *const rxr_pattern,
    pub n_patterns: u32,
    pub normal_patterns: *const rxr_pattern,
    pub n_normal_patterns: u32,
    pub palette_rgb: *const f32,
    pub palette_present: *const u8,
    pub n_palette: u32,
}

extern "C" {
    pub fn rxr_create(out: *mut *mut rxr_ctx, device_id: c_int) -> c_int;
    pub fn rxr_destroy(ctx: *mut rxr_ctx);
    pub fn rxr_last_error(ctx: *const rxr_ctx) -> *const c_char;
    pub fn rxr_device_count() -> c_int;
    pub fn rxr_set_textures(ctx: *mut rxr_ctx, s: *const rxr_tile, ns: u32, d: *const rxr_tile, nd: u32) -> c_int;
    pub fn rxr_set_meshes(ctx: *mut rxr_ctx, meshes: *const rxr_mesh3d, n: u32) -> c_int;
    pub fn rxr_set_shaders(ctx: *mut rxr_ctx, set: *const rxr_shader_set) -> c_int;
    pub fn rxr_upload_frame(ctx: *mut rxr_ctx, frame: *const rxr_frame) -> c_int;
    pub fn rxr_render_rows(ctx: *mut rxr_ctx, row0: u32, row1: u32) -> c_int;
    pub fn rxr_render_rows_to(ctx: *mut rxr_ctx, row0: u32, row1: u32, dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rxr_render_stripes_to(ctx: *mut rxr_ctx, first: u32, stride: u32, dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rxr_download_rows(ctx: *mut rxr_ctx, pixels: *mut u8, row0: u32, row1: u32) -> c_int;
    pub fn rxr_rasterize(ctx: *mut rxr_ctx, frame: *const rxr_frame, pixels: *mut u8) -> c_int;
    pub fn rxr_render_download(ctx: *mut rxr_ctx, pixels: *mut u8) -> c_int;
    pub fn rxr_synchronize(ctx: *mut rxr_ctx) -> c_int;
}
