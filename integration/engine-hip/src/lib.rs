//! `impl Renderer` / `impl FrameIterator` over librenderbaby_hip.so -- the crate RenderBaby adds
//! to select the HIP backend.  Source only -- not compiled here.
mod ffi;
use anyhow::{anyhow, Result};
use engine_config::{render_config::Change, RenderConfig, Renderer};
use frame_buffer::frame_iterator::{Frame, FrameIterator};
use std::{ffi::CStr, os::raw::c_void, ptr, sync::{Arc, Mutex}};

struct Handle(*mut ffi::RbEngine);
unsafe impl Send for Handle {}          // the library serialises every call internally and
unsafe impl Sync for Handle {}          // selects its device per call (any thread may call)
impl Drop for Handle { fn drop(&mut self) { unsafe { ffi::rb_destroy(self.0) } } }

pub struct Engine { h: Arc<Mutex<Handle>> }   // mirrors Arc<Mutex<GpuWrapper>>, lib.rs:39-42

fn field<T>(c: &Change<Vec<T>>) -> ffi::RbField {
    match c {
        Change::Keep      => ffi::RbField { change: ffi::RB_KEEP,   ptr: ptr::null(), count: 0 },
        Change::Delete    => ffi::RbField { change: ffi::RB_DELETE, ptr: ptr::null(), count: 0 },
        Change::Create(v) => ffi::RbField { change: ffi::RB_CREATE, ptr: v.as_ptr() as *const c_void, count: v.len() },
        Change::Update(v) => ffi::RbField { change: ffi::RB_UPDATE, ptr: v.as_ptr() as *const c_void, count: v.len() },
    }
}

/// Borrow `rc` as an `rb_config` for the duration of `f` (the library copies to the device
/// before it returns; nothing is retained).
fn with_config<R>(rc: &RenderConfig, f: impl FnOnce(&ffi::RbConfig) -> R) -> R {
    let uniforms = match &rc.uniforms {
        Change::Keep      => ffi::RbField { change: ffi::RB_KEEP,   ptr: ptr::null(), count: 0 },
        Change::Delete    => ffi::RbField { change: ffi::RB_DELETE, ptr: ptr::null(), count: 0 },
        Change::Create(u) => ffi::RbField { change: ffi::RB_CREATE, ptr: u as *const _ as *const c_void, count: 1 },
        Change::Update(u) => ffi::RbField { change: ffi::RB_UPDATE, ptr: u as *const _ as *const c_void, count: 1 },
    };
    let (tex_change, tex): (u32, Vec<ffi::RbTexture>) = match &rc.textures {
        Change::Keep => (ffi::RB_KEEP, vec![]), Change::Delete => (ffi::RB_DELETE, vec![]),
        Change::Create(t) | Change::Update(t) => (
            if matches!(rc.textures, Change::Create(_)) { ffi::RB_CREATE } else { ffi::RB_UPDATE },
            t.iter().map(|t| ffi::RbTexture { width: t.width, height: t.height, rgba_data: t.rgba_data.as_ptr() }).collect()),
    };
    let cfg = ffi::RbConfig {
        uniforms, spheres: field(&rc.spheres), uvs: field(&rc.uvs), meshes: field(&rc.meshes),
        lights: field(&rc.lights), bvh_nodes: field(&rc.bvh_nodes), bvh_indices: field(&rc.bvh_indices),
        bvh_triangles: field(&rc.bvh_triangles),
        textures: ffi::RbField { change: tex_change, ptr: tex.as_ptr() as *const c_void, count: tex.len() },
    };
    f(&cfg)
}

fn check(e: *mut ffi::RbEngine, rc: i32) -> Result<()> {
    if rc == 0 { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(ffi::rb_last_error(e)) }.to_string_lossy().into_owned();
    Err(anyhow!(msg))                      // e.g. "Invalid Spheres", "No more frames available"
}

fn frame_for(e: *mut ffi::RbEngine) -> Result<Frame> {
    let (mut w, mut h) = (0u32, 0u32);
    check(e, unsafe { ffi::rb_get_size(e, &mut w, &mut h) })?;
    Ok(Frame::new(w as usize, h as usize, vec![0u8; w as usize * h as usize * 4]))
}

/// What has no counterpart in the reference (one wgpu device, one pass per frame, its own walk): which devices,
/// which walk for multi-node meshes, how often the iterator delivers.  `Default` = one device, the reference's
/// behaviour throughout.
#[derive(Default, Clone)]
pub struct EngineOptions {
    pub devices: Vec<i32>,            // empty: the current device; several: one handle over all of them
    pub stripe_rows: u32,             // rows per stripe of the row sharding (0 = 8)
    pub own_tree: bool,               // RB_FLAG_FAST_BVH: the library's tree for multi-node meshes (same frames)
    pub device_built_tree: bool,      // RB_FLAG_DEVICE_BVH
    pub passes_per_frame: u32,        // iterator: a frame every n samples (0 / 1 = every sample)
}

impl Engine {
    pub fn new(rc: RenderConfig) -> Self {                       // engine-pathtracer lib.rs:111-119
        Self::with_options(rc, &EngineOptions::default())
    }
    pub fn with_options(rc: RenderConfig, o: &EngineOptions) -> Self {
        let opt = ffi::RbOptions {
            device: -1, stripe_rows: o.stripe_rows,
            flags: if o.own_tree { ffi::RB_FLAG_FAST_BVH } else { 0 } | if o.device_built_tree { ffi::RB_FLAG_DEVICE_BVH } else { 0 },
            ..Default::default()
        };
        let e = with_config(&rc, |c| unsafe {
            if o.devices.len() > 1 {
                // rows in interleaved stripes over the devices, ONE RCCL gather per delivered frame inside the library
                ffi::rb_create_multi(c, &opt, o.devices.as_ptr(), o.devices.len() as u32)
            } else {
                let opt = ffi::RbOptions { device: o.devices.first().copied().unwrap_or(-1), ..opt };
                ffi::rb_create_ex(c, &opt)
            }
        });
        assert!(!e.is_null(), "{}", unsafe { CStr::from_ptr(ffi::rb_last_error(ptr::null())) }.to_string_lossy());
        if o.passes_per_frame > 1 { unsafe { ffi::rb_iter_set_passes_per_frame(e, o.passes_per_frame); } }
        Self { h: Arc::new(Mutex::new(Handle(e))) }
    }
}

impl Renderer for Engine {
    fn render(&mut self, rc: RenderConfig) -> Result<Frame> {   // lib.rs:58-71
        let g = self.h.lock().unwrap();
        check(g.0, with_config(&rc, |c| unsafe { ffi::rb_update(g.0, c) }))?;
        let mut f = frame_for(g.0)?;
        check(g.0, unsafe { ffi::rb_render(g.0, f.pixels.as_mut_ptr()) })?;
        Ok(f)
    }
    fn frame_iterator(&mut self, rc: RenderConfig) -> Result<Box<dyn FrameIterator>> {   // lib.rs:86-96
        let g = self.h.lock().unwrap();
        check(g.0, with_config(&rc, |c| unsafe { ffi::rb_iter_begin(g.0, c) }))?;
        Ok(Box::new(HipFrameIterator { h: Arc::clone(&self.h) }))
    }
}

pub struct HipFrameIterator { h: Arc<Mutex<Handle>> }

impl FrameIterator for HipFrameIterator {
    fn has_next(&self) -> bool { unsafe { ffi::rb_iter_has_next(self.h.lock().unwrap().0) != 0 } }
    fn next(&mut self) -> Result<Frame> {                        // lib.rs:169-228
        let g = self.h.lock().unwrap();
        let mut f = frame_for(g.0)?;
        check(g.0, unsafe { ffi::rb_iter_next(g.0, f.pixels.as_mut_ptr()) })?;
        Ok(f)
    }
    fn destroy(&mut self) { unsafe { ffi::rb_iter_destroy(self.h.lock().unwrap().0) } }
}
