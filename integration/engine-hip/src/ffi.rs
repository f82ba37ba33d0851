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
pub const RB_FLAG_FAST_BVH: u32 = 4;             // multi-node meshes: the library's own tree (argued and fuzzed exact, two passes)
pub const RB_FLAG_DEVICE_BVH: u32 = 8;           // build that tree on the GPU
pub const RB_FLAG_REFERENCE_WALK: u32 = 32;      // force the reference's walk (no flag = the chunked walk, RB_FLAG_CHUNK_WALK = 1024)
pub const RB_FLAG_HOST_BVH: u32 = 64;            // build that tree on the host
pub const RB_FLAG_GATHER_PEER_COPY: u32 = 128;   // rb_create_multi without RCCL
pub const RB_FLAG_NO_RUN_AHEAD: u32 = 256;       // iterator: no pass started ahead of the read-back
pub const RB_FLAG_SKIP_NEAR_DEGENERATE: u32 = 512; // the library's tree without its second pass (outside the exactness argument)
pub const RB_FLAG_CHUNK_WALK: u32 = 1024;        // the chunked walk: the default for multi-node meshes, the flag only names it
pub const RB_FLAG_SPHERE_TREE_HOST: u32 = 2048;  // > 64 spheres: build the sphere tree on the host ...
pub const RB_FLAG_SPHERE_TREE_DEVICE: u32 = 4096; // ... or on the device whatever the count (default: the device from 1024 spheres up)
pub const RB_FLAG_CHUNK_TREE_HOST: u32 = 8192;    // the chunked walk's tree: built on the host ...
pub const RB_FLAG_CHUNK_TREE_DEVICE: u32 = 16384; // ... or on the device whatever the size (default: the device from 16 384 triangle slots up)
pub const RB_COMM_ID_BYTES: usize = 128;

unsafe extern "C" {
    pub fn rb_create(cfg: *const RbConfig) -> *mut RbEngine;
    pub fn rb_create_ex(cfg: *const RbConfig, opt: *const RbOptions) -> *mut RbEngine;
    /// one handle over several devices of this process: rows sharded in stripes, one RCCL gather per delivered frame
    pub fn rb_create_multi(cfg: *const RbConfig, opt: *const RbOptions, devices: *const i32, n_devices: u32) -> *mut RbEngine;
    /// one process per device: every rank asks whether RCCL loads at all (no id, no socket), rank 0 makes the id, every rank joins with its shard
    pub fn rb_comm_available() -> c_int;
    pub fn rb_comm_unique_id(id_out: *mut u8) -> c_int;
    pub fn rb_comm_init_rank(e: *mut RbEngine, id: *const u8, rank: u32, nranks: u32) -> c_int;
    /// the communicator as RCCL reports it (0 ranks: nothing goes through RCCL) and this rank's share of the last gather
    pub fn rb_comm_info(e: *mut RbEngine, rccl_ranks: *mut u32, rccl_rank: *mut u32, last_gather_ms: *mut f32) -> c_int;
    /// page-locked frame memory: read-backs into it are DMA copies that overlap the next pass
    pub fn rb_host_alloc(bytes: usize) -> *mut c_void;
    pub fn rb_host_free(p: *mut c_void);
    pub fn rb_destroy(e: *mut RbEngine);
    pub fn rb_update(e: *mut RbEngine, cfg: *const RbConfig) -> c_int;
    pub fn rb_render(e: *mut RbEngine, rgba_out: *mut u8) -> c_int;
    pub fn rb_iter_begin(e: *mut RbEngine, cfg: *const RbConfig) -> c_int;
    pub fn rb_iter_has_next(e: *mut RbEngine) -> c_int;
    pub fn rb_iter_next(e: *mut RbEngine, rgba_out: *mut u8) -> c_int;
    pub fn rb_iter_destroy(e: *mut RbEngine);
    pub fn rb_iter_set_passes_per_frame(e: *mut RbEngine, n: u32) -> c_int;   // extension: a frame every n samples
    /// optional: allocate what the first render of n_passes passes would allocate lazily (a host that times its first frame)
    pub fn rb_reserve(e: *mut RbEngine, n_passes: u32) -> c_int;
    pub fn rb_get_size(e: *const RbEngine, w: *mut u32, h: *mut u32) -> c_int;
    /// BVH::new (engine-bvh/src/bvh.rs:87-150) restated, its top levels forked onto threads: the tree the adapter rebuilds per render
    /// (scene_engine_adapter.rs:435-440) in a tenth of the time.  Two calls: nodes_out = null asks for the sizes.
    pub fn rb_bvh_build(tris: *const c_void, n_tris: usize, nodes_out: *mut c_void, nodes_capacity: usize, n_nodes: *mut usize, indices_out: *mut u32) -> c_int;
    pub fn rb_last_error(e: *const RbEngine) -> *const c_char;
}
