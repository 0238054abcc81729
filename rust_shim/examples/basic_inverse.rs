// UNVERIFIED (never compiled here).  The reference's src/examples/basic_inverse.rs test body (:130-258) against this crate:
// Inverse (1/n fused), n = 512, constant input 2 + 42i, result copied to a staging buffer and mapped, max abs error < 1e-5.
// The compiled equivalent that the tests run is tools/example_basic_inverse.cpp (same call sequence through the C ABI).
use fft_wgpu::{wgpu, Complex, Inverse};

fn main() {
    let (device, queue) = fft_wgpu::prepare_gpu().expect("no gfx950 device");
    let n = 512usize;
    let count = n * 500 * 5;
    let data = vec![Complex::new(2.0, 42.0); count];
    let bytes = (count * 8) as u64;
    let staging = device.create_buffer(&wgpu::BufferDescriptor { label: None, size: bytes, usage: wgpu::BufferUsages::MAP_READ | wgpu::BufferUsages::COPY_DST, mapped_at_creation: false });
    let src = device.create_buffer(&wgpu::BufferDescriptor {
        label: None,
        size: bytes,
        usage: wgpu::BufferUsages::COPY_DST | wgpu::BufferUsages::COPY_SRC | wgpu::BufferUsages::STORAGE,
        mapped_at_creation: false,
    });
    let fft_inverse = Inverse::new(&device, &queue, &src, n as u32);
    queue.write_buffer(&src, 0, bytemuck::cast_slice(&data));
    let mut encoder = device.create_command_encoder(&wgpu::CommandEncoderDescriptor { label: None });
    let output = fft_inverse.proc(&mut encoder); // log2 512 is odd: the plan's second buffer (processor.rs:335-339)
    encoder.copy_buffer_to_buffer(output, 0, &staging, 0, bytes);
    queue.submit(Some(encoder.finish()));
    let slice = staging.slice(..);
    slice.map_async(wgpu::MapMode::Read, |_| {});
    device.poll(wgpu::Maintain::wait()).panic_on_timeout();
    let mut ans = vec![Complex::zero(); count];
    {
        let view = slice.get_mapped_range();
        ans.copy_from_slice(bytemuck::cast_slice(&view));
    }
    staging.unmap();
    let mut worst = 0f32;
    for (i, v) in ans.iter().enumerate() {
        let (er, ei) = if i % n == 0 { (2.0, 42.0) } else { (0.0, 0.0) };
        worst = worst.max((v.real - er).abs()).max((v.imag - ei).abs());
    }
    assert!(worst < 1e-5, "max error {worst}");
    println!("max error {worst}");
}
