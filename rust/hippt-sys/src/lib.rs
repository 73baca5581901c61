//! Raw bindings to include/hippt.h (the subset the `--gpu` path needs).
//! UNVERIFIED IN THIS ENVIRONMENT (no Rust toolchain in the build image).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int};

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct hippt_material { pub rgb: [f32; 3], pub metallic: f32, pub roughness: f32, pub ior: f32,
                            pub volume_density: f32, pub volume_anisotropy: f32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct hippt_sphere { pub center: [f32; 3], pub radius: f32, pub material: u32, pub order: u32, pub in_focus: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct hippt_plane { pub point: [f32; 3], pub u: [f32; 3], pub v: [f32; 3], pub normal: [f32; 3],
                         pub material: u32, pub order: u32, pub in_focus: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct hippt_light { pub pos: [f32; 3], pub u: [f32; 3], pub v: [f32; 3], pub intensity: [f32; 3] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct hippt_camera { pub pos: [f32; 3], pub right: [f32; 3], pub up: [f32; 3], pub forward: [f32; 3],
                          pub aspect: f32, pub scale: f32, pub aperture: f32, pub focus: f32 }
#[repr(C)]
pub struct hippt_scene {
    pub camera: hippt_camera,
    pub width: u32, pub height: u32, pub samples: u32,
    pub n_materials: u32, pub n_spheres: u32, pub n_planes: u32, pub n_lights: u32, pub reserved: u32,
    pub materials: *const hippt_material, pub spheres: *const hippt_sphere,
    pub planes: *const hippt_plane, pub lights: *const hippt_light,
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct hippt_options { pub sample_begin: u32, pub sample_count: u32, pub max_depth: u32, pub seed: u32,
                           pub flags: u32, pub device: i32 }

// hippt_options.flags (include/hippt.h)
pub const HIPPT_FLAG_LOCAL_REDUCE: u32 = 128;     // *_multi: a device may appear several times, device add instead of ncclReduce
pub const HIPPT_FLAG_COUNT_EXECUTED: u32 = 512;  // with COUNTERS: shadow tests / evaluations as the timed kernel executes them
pub const HIPPT_FLAG_TRAVERSAL_STATS: u32 = 256;  // LBVH pipeline: count node / leaf fetches (hippt_read_traversal_stats)

extern "C" {
    pub fn hippt_last_error() -> *const c_char;
    pub fn hippt_render_rgba8(scene: *const hippt_scene, opts: *const hippt_options, out_rgba: *mut u8) -> c_int;
    pub fn hippt_render_accum_f32(scene: *const hippt_scene, opts: *const hippt_options, out_accum: *mut f32,
                                  counters: *mut u64) -> c_int;
    // one process, several GPUs: sample-range split + one RCCL reduce inside the library
    pub fn hippt_render_rgba8_multi(scene: *const hippt_scene, opts: *const hippt_options, devices: *const c_int,
                                    n_devices: c_int, out_rgba: *mut u8) -> c_int;
    pub fn hippt_render_accum_f32_multi(scene: *const hippt_scene, opts: *const hippt_options, devices: *const c_int,
                                        n_devices: c_int, out_accum: *mut f32) -> c_int;
    pub fn hippt_sample_range(total: u32, rank: u32, world: u32, begin: *mut u32, count: *mut u32);
    pub fn hippt_device_count(count: *mut c_int) -> c_int;
    // resident context + progressive accumulation (long jobs, checkpoints)
    pub fn hippt_create(device: c_int, out_ctx: *mut *mut hippt_context) -> c_int;
    pub fn hippt_destroy(ctx: *mut hippt_context) -> c_int;
    pub fn hippt_set_scene(ctx: *mut hippt_context, scene: *const hippt_scene) -> c_int;
    pub fn hippt_render_accum_add_f32(ctx: *mut hippt_context, opts: *const hippt_options, host_accum: *mut f32) -> c_int;
    pub fn hippt_resolve_f32(ctx: *mut hippt_context, host_accum: *const f32, n_pixels: u32, spp: u32,
                             out_rgba: *mut u8) -> c_int;
    pub fn hippt_checkpoint_save(path: *const c_char, accum: *const f32, width: u32, height: u32,
                                 samples_done: u32, seed: u32) -> c_int;
    // device-resident rendering on a caller's HIP stream (accumulators in device memory), as bench.py drives it
    pub fn hippt_render_accum_device(ctx: *mut hippt_context, opts: *const hippt_options, d_accum: *mut f32,
                                     stream: *mut std::os::raw::c_void) -> c_int;
    // the same for callers that cannot allocate (HIP graph capture, serving loops): table scenes, <= 4 194 304 pixels
    pub fn hippt_render_accum_device_noalloc(ctx: *mut hippt_context, opts: *const hippt_options, d_accum: *mut f32,
                                             stream: *mut std::os::raw::c_void) -> c_int;
    pub fn hippt_resolve_device(ctx: *mut hippt_context, d_accum: *const f32, spp: u32, d_rgba: *mut u8,
                                stream: *mut std::os::raw::c_void) -> c_int;
    pub fn hippt_kernel_time_ms(ctx: *mut hippt_context, total_ms: *mut f64, launches: *mut u32) -> c_int;
    pub fn hippt_read_traversal_stats(ctx: *mut hippt_context, stats10: *mut u64) -> c_int;
    pub fn hippt_checkpoint_load(path: *const c_char, accum: *mut f32, width: u32, height: u32,
                                 samples_done: *mut u32, seed: *mut u32) -> c_int;
}
#[repr(C)] pub struct hippt_context { _private: [u8; 0] }
