// UNVERIFIED (never compiled here: no Rust toolchain).  A forward-transform round trip written against this crate with the
// `wgpu` names a caller of the reference crate uses (the reference's own example, src/examples/basic.rs:3-131, needs only
// its `wgpu` import pointed at `fft_wgpu::wgpu` to build against this crate): adapter / device futures, usage flags, one
// encoder per iteration, `slice(..)` + `map_async` + `poll(wait)` + `get_mapped_range` + `unmap` for the read-back.
// Unlike the reference's example this one checks its result.  Its compiled twin is tools/example_basic_pipeline.cpp.
use fft_wgpu::wgpu;
use fft_wgpu::{Complex, Forward};

const FFT_LEN: u32 = 512;
const TRANSFORMS: usize = 2500;
const ROUNDS: usize = 100;

fn device_buffer(device: &wgpu::Device, bytes: u64, usage: wgpu::BufferUsages) -> wgpu::Buffer {
    device.create_buffer(&wgpu::BufferDescriptor { label: None, size: bytes, usage, mapped_at_creation: false })
}

async fn open_gpu() -> (wgpu::Device, wgpu::Queue) {
    let instance = wgpu::Instance::default();
    let options = wgpu::RequestAdapterOptions { power_preference: wgpu::PowerPreference::HighPerformance, ..Default::default() };
    let adapter = instance.request_adapter(&options).await.expect("no gfx950 adapter");
    let wanted = wgpu::DeviceDescriptor { required_features: adapter.features(), required_limits: adapter.limits(), ..Default::default() };
    adapter.request_device(&wanted, None).await.expect("device")
}

#[tokio::main]
async fn main() {
    let (device, queue) = open_gpu().await;
    // impulse at position 3 of every transform: X[k] = exp(-2 pi i 3 k / n)
    let samples = FFT_LEN as usize * TRANSFORMS;
    let mut input = vec![Complex::zero(); samples];
    for t in 0..TRANSFORMS {
        input[t * FFT_LEN as usize + 3] = Complex::new(1.0, 0.0);
    }
    let bytes = (samples * std::mem::size_of::<Complex>()) as u64;
    let readback = device_buffer(&device, bytes, wgpu::BufferUsages::MAP_READ | wgpu::BufferUsages::COPY_DST);
    let signal = device_buffer(&device, bytes, wgpu::BufferUsages::STORAGE | wgpu::BufferUsages::COPY_SRC | wgpu::BufferUsages::COPY_DST);
    let plan = Forward::new(&device, &queue, &signal, FFT_LEN);
    let whole = readback.slice(..);
    let mut spectrum = vec![Complex::zero(); samples];

    let started = std::time::Instant::now();
    for _ in 0..ROUNDS {
        queue.write_buffer(&signal, 0, bytemuck::cast_slice(&input));
        let mut encoder = device.create_command_encoder(&wgpu::CommandEncoderDescriptor { label: None });
        let result = plan.proc(&mut encoder);
        encoder.copy_buffer_to_buffer(result, 0, &readback, 0, bytes);
        queue.submit(Some(encoder.finish()));
        whole.map_async(wgpu::MapMode::Read, |outcome| outcome.expect("map"));
        device.poll(wgpu::Maintain::wait()).panic_on_timeout();
        {
            let view = whole.get_mapped_range();
            spectrum.copy_from_slice(bytemuck::cast_slice(&view));
        }
        readback.unmap();
    }
    let mut worst = 0f32;
    for (i, v) in spectrum.iter().enumerate() {
        let k = (i % FFT_LEN as usize) as f32;
        let phase = -2.0 * std::f32::consts::PI * 3.0 * k / FFT_LEN as f32;
        worst = worst.max((v.real - phase.cos()).abs()).max((v.imag - phase.sin()).abs());
    }
    println!("{ROUNDS} rounds of {TRANSFORMS} x {FFT_LEN} in {:?}, max abs error {worst:e}", started.elapsed());
    assert!(worst < 1e-5);
}
