// UNVERIFIED (never compiled here).  The call sequence of the reference's forward benchmark, src/examples/basic.rs:3-131,
// written against this crate: the ONLY change a maintainer makes to the reference file is the import below (the `wgpu`
// names resolve to fft_wgpu::wgpu_helper, whose objects own C-ABI handles of include/fft_wgpu_amd.h).  Every wgpu item the
// reference example touches appears here once: Instance, request_adapter / request_device futures, BufferUsages flags,
// slice(..) + map_async + poll(wait) + get_mapped_range + unmap.  The compiled equivalent the tests run is
// tools/example_basic_pipeline.cpp (same loop through the C ABI, every read-back sample checked).
use fft_wgpu::wgpu;
use fft_wgpu::Complex;

#[tokio::main]
async fn main() {
    let instance = wgpu::Instance::default();
    let adapter = instance
        .request_adapter(&wgpu::RequestAdapterOptions { power_preference: wgpu::PowerPreference::HighPerformance, ..Default::default() })
        .await
        .unwrap();
    let (device, queue) = adapter
        .request_device(
            &wgpu::DeviceDescriptor { required_features: adapter.features(), required_limits: adapter.limits(), ..Default::default() },
            None,
        )
        .await
        .unwrap();

    let fft_len = 512u32;
    let data = vec![Complex::new(1.0, 0.0); 512 * 2500];
    let bytes = (data.len() * std::mem::size_of::<Complex>()) as u64;
    let mut ans = vec![Complex::zero(); data.len()];
    let staging_buffer = device.create_buffer(&wgpu::BufferDescriptor {
        label: None,
        size: bytes,
        usage: wgpu::BufferUsages::MAP_READ | wgpu::BufferUsages::COPY_DST,
        mapped_at_creation: false,
    });
    let src = device.create_buffer(&wgpu::BufferDescriptor {
        label: None,
        size: bytes,
        usage: wgpu::BufferUsages::COPY_DST | wgpu::BufferUsages::COPY_SRC | wgpu::BufferUsages::STORAGE,
        mapped_at_creation: false,
    });
    let fft_forward = fft_wgpu::Forward::new(&device, &queue, &src, fft_len);
    let buffer_slice = staging_buffer.slice(..);

    let timer = std::time::Instant::now();
    for _ in 0..1000 {
        queue.write_buffer(&src, 0, bytemuck::cast_slice(data.as_slice()));
        let mut encoder = device.create_command_encoder(&wgpu::CommandEncoderDescriptor { label: None });
        let output = fft_forward.proc(&mut encoder);
        encoder.copy_buffer_to_buffer(output, 0, &staging_buffer, 0, bytes);
        queue.submit(Some(encoder.finish()));
        buffer_slice.map_async(wgpu::MapMode::Read, move |_| {});
        device.poll(wgpu::Maintain::wait()).panic_on_timeout();
        let mapped = buffer_slice.get_mapped_range();
        ans.copy_from_slice(bytemuck::cast_slice(&mapped));
        drop(mapped);
        staging_buffer.unmap();
    }
    println!("1000 iterations in {:?}; X[0] of the first transform = {:?} (expected 512 + 0i)", timer.elapsed(), ans[0]);
    assert!((ans[0].real - 512.0).abs() < 1e-3 && ans[1].real.abs() < 1e-4);
}
