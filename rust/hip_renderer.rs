// src/hip_renderer.rs — drop-in for src/gpu_renderer.rs of eliasstepanik/Pathtracer.
//
//     pub fn render(scene: &Scene) -> RgbaImage
//
// same signature, same output (RGBA8, alpha 255, row 0 = top, ACES + gamma 2.2), same blocking
// single-call shape; main.rs only changes `gpu_renderer::render` to `hip_renderer::render` (main.rs:91)
// and adds `mod hip_renderer;`. All rendering happens in the HIP kernels of libhippt through hippt-sys;
// there is no wgpu fallback inside this path.
// UNVERIFIED IN THIS ENVIRONMENT (no Rust toolchain in the build image); the identical marshalling is
// exercised from Python (pathtracer_amd/scene.py, hip_renderer.py) and C++ (hippt_cli.cpp).
use crate::{object::Object, scene::Scene};
use hippt_sys::*;
use image::RgbaImage;
use std::ffi::CStr;

pub fn render(scene: &Scene) -> RgbaImage {
    let (width, height) = (scene.render.width, scene.render.height);

    // camera basis and autofocus exactly as main.rs:44-56
    let aspect = width as f32 / height as f32;
    let scale = (scene.camera.fov.to_radians() * 0.5).tan();
    let forward = (scene.camera.look_at - scene.camera.pos).normalize();
    let right = scene.camera.up.cross(forward).normalize();
    let up = forward.cross(right).normalize();
    let focus = crate::renderer::autofocus(scene.camera.pos, right, up, forward, aspect, scale, width, height,
                                           &scene.objects);
    let v3 = |v: crate::algebra::Vec3| [v.0, v.1, v.2];

    // one material per object (the loader has already resolved names, scene.rs:128,138)
    let mut materials = Vec::new();
    let mut spheres = Vec::new();
    let mut planes = Vec::new();
    for (order, obj) in scene.objects.iter().enumerate() {
        let m = match obj { Object::Sphere(s) => s.material, Object::Plane(p) => p.material };
        materials.push(hippt_material { rgb: v3(m.color), metallic: m.metallic, roughness: m.roughness, ior: m.ior,
                                        volume_density: m.volume_density, volume_anisotropy: m.volume_anisotropy });
        let material = (materials.len() - 1) as u32;
        match obj {
            Object::Sphere(s) => spheres.push(hippt_sphere { center: v3(s.center), radius: s.radius, material,
                                                             order: order as u32, in_focus: s.in_focus as u32 }),
            Object::Plane(p) => planes.push(hippt_plane { point: v3(p.point), u: v3(p.u), v: v3(p.v), normal: v3(p.normal),
                                                          material, order: order as u32, in_focus: p.in_focus as u32 }),
        }
    }
    let lights: Vec<hippt_light> = scene.lights.iter()
        .map(|l| hippt_light { pos: v3(l.pos), u: v3(l.u), v: v3(l.v), intensity: v3(l.intensity) }).collect();

    let c_scene = hippt_scene {
        camera: hippt_camera { pos: v3(scene.camera.pos), right: v3(right), up: v3(up), forward: v3(forward),
                               aspect, scale, aperture: scene.camera.aperture, focus },
        width, height, samples: scene.render.samples,
        n_materials: materials.len() as u32, n_spheres: spheres.len() as u32, n_planes: planes.len() as u32,
        n_lights: lights.len() as u32, reserved: 0,
        materials: materials.as_ptr(), spheres: spheres.as_ptr(), planes: planes.as_ptr(), lights: lights.as_ptr(),
    };
    let opts = hippt_options { seed: rand::random(), ..Default::default() };
    let mut rgba = vec![0u8; (width * height * 4) as usize];
    // HIPPT_DEVICES=0,1,...,7 (or `all`) renders on several GPUs of the node (sample-range split, RCCL reduce in the library)
    let devices: Vec<i32> = match std::env::var("HIPPT_DEVICES").ok().as_deref() {
        Some("all") => { let mut n = 0; unsafe { hippt_device_count(&mut n) }; (0..n).collect() }
        Some(s) => s.split(',').filter_map(|d| d.trim().parse().ok()).collect(),
        None => Vec::new(),
    };
    let rc = if devices.is_empty() {
        unsafe { hippt_render_rgba8(&c_scene, &opts, rgba.as_mut_ptr()) }
    } else {
        unsafe { hippt_render_rgba8_multi(&c_scene, &opts, devices.as_ptr(), devices.len() as i32, rgba.as_mut_ptr()) }
    };
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(hippt_last_error()) }.to_string_lossy().into_owned();
        panic!("HIP renderer failed ({rc}): {msg}");       // same abort-with-message as gpu_renderer's `expect`s
    }
    RgbaImage::from_raw(width, height, rgba).expect("image buffer size")
}
