//! 1:1 binding of include/rb_abi.h (the part the Renderer / FrameIterator shim needs).
//! Source only -- not compiled in the renderbaby-hip repository (no Rust toolchain there).
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct RbField { pub change: u32, pub ptr: *const c_void, pub count: usize }
#[repr(C)] pub struct RbConfig {
    pub uniforms: RbField, pub spheres: RbField, pub uvs: RbField, pub meshes: RbField,
    pub lights: RbField, pub bvh_nodes: RbField, pub bvh_indices: RbField,
    pub bvh_triangles: RbField, pub textures: RbField,
}
#[repr(C)] pub struct RbTexture { pub width: u32, pub height: u32, pub rgba_data: *const u32 }
#[repr(C)] pub struct RbEngine { _private: [u8; 0] }

pub const RB_KEEP: u32 = 0; pub const RB_CREATE: u32 = 1;
pub const RB_UPDATE: u32 = 2; pub const RB_DELETE: u32 = 3;

/// rb_options (48 bytes): device, row sharding, launch chunking, kernel choice, opt-in flags.
#[repr(C)] #[derive(Default, Clone, Copy)]
pub struct RbOptions {
    pub device: i32, pub shard_rank: u32, pub shard_count: u32, pub stripe_rows: u32,
    pub passes_per_launch: u32, pub kernel: u32, pub flags: u32, pub reserved: [u32; 5],
}
pub const RB_FLAG_FAST_BVH: u32 = 4;     // opt-in small-leaf tree with culling, same frames
pub const RB_FLAG_DEVICE_BVH: u32 = 8;   // build that tree on the GPU

unsafe extern "C" {
    pub fn rb_create(cfg: *const RbConfig) -> *mut RbEngine;
    pub fn rb_create_ex(cfg: *const RbConfig, opt: *const RbOptions) -> *mut RbEngine;
    pub fn rb_destroy(e: *mut RbEngine);
    pub fn rb_update(e: *mut RbEngine, cfg: *const RbConfig) -> c_int;
    pub fn rb_render(e: *mut RbEngine, rgba_out: *mut u8) -> c_int;
    pub fn rb_iter_begin(e: *mut RbEngine, cfg: *const RbConfig) -> c_int;
    pub fn rb_iter_has_next(e: *mut RbEngine) -> c_int;
    pub fn rb_iter_next(e: *mut RbEngine, rgba_out: *mut u8) -> c_int;
    pub fn rb_iter_destroy(e: *mut RbEngine);
    pub fn rb_iter_set_passes_per_frame(e: *mut RbEngine, n: u32) -> c_int;   // extension: a frame every n samples
    pub fn rb_get_size(e: *const RbEngine, w: *mut u32, h: *mut u32) -> c_int;
    pub fn rb_last_error(e: *const RbEngine) -> *const c_char;
}
