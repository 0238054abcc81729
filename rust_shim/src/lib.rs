//! fft_wgpu (MI355X build) -- UNVERIFIED SOURCE: never compiled in this pipeline (no Rust toolchain in the image or on
//! the GPU box).  Same public surface as the reference crate's `src/lib.rs` (:6-27): `Complex`, `pub use processor::*`,
//! `pub mod wgpu_helper` -- but `wgpu_helper`, an empty file in the reference (`src/lib.rs:8`), is where the GPU
//! layer now lives: the handful of `wgpu` objects the plans and the examples touch, implemented over the C ABI of
//! `include/fft_wgpu_amd.h` (hand-written HIP for gfx950).
//!
//! Switching a caller over: replace the `wgpu` dependency by this crate's re-export,
//! `use fft_wgpu::wgpu;` (or `use fft_wgpu::wgpu_helper as wgpu;`).  `Forward::new(&device, &queue, &src, fft_len)`,
//! `proc(&self, &mut encoder) -> &Buffer` and the other three plans keep the reference's signatures
//! (`src/processor.rs:22-27,110,245-250,293,422-428,467,580-586,622`).
use bytemuck::{Pod, Zeroable};

pub mod ffi;
pub mod processor;
pub use processor::*;
pub mod wgpu_helper;
pub mod sharded;
/// Drop-in name for the objects the reference takes from the `wgpu` crate.
pub use wgpu_helper as wgpu;

/// Wire layout of every buffer: interleaved re, im, little-endian f32 (reference `src/lib.rs:10-15`).
#[repr(C)]
#[derive(Copy, Clone, Debug, Pod, Zeroable)]
pub struct Complex {
    pub real: f32,
    pub imag: f32,
}

impl Complex {
    pub fn new(re: f32, im: f32) -> Self {
        Self { real: re, imag: im }
    }
    pub fn zero() -> Self {
        Self { real: 0.0, imag: 0.0 }
    }
}

/// Reference `src/lib.rs:29-62` (`prepare_gpu`): lists the adapters (`:33-35`), asks for a high-performance one, opens a
/// device and its queue; `None` when no usable GPU exists.
pub fn prepare_gpu() -> Option<(wgpu::Device, wgpu::Queue)> {
    let instance = wgpu::Instance::default();
    for adapter in instance.enumerate_adapters(wgpu::Backends::all()) {
        eprintln!("{:?}", adapter.get_info()); // the reference `dbg!`s its adapter lists
    }
    let adapter = instance.enumerate_adapters(wgpu::Backends::all()).into_iter().next()?;
    wgpu::Device::open(adapter.ordinal()).map(|d| {
        let q = d.queue();
        (d, q)
    })
}
