// UNVERIFIED (never compiled here).  The reference's src/examples/basic_inverse2.rs test body (:139-286) against this
// crate: Onlyinverse + Normalize in one encoder, n = 512, constant input, max abs error < 1e-5.  The compiled
// equivalent that the tests actually run is tools/example_basic_inverse2.cpp (same call sequence through the C ABI).
use fft_wgpu::{wgpu, Complex, Normalize, Onlyinverse};

fn main() {
    let (device, queue) = fft_wgpu::prepare_gpu().expect("no gfx950 device");
    let n = 512usize;
    let count = n * 500 * 5;
    let data = vec![Complex::new(2.1327392395, 3.033729); count];
    let bytes = (count * 8) as u64;
    let desc = wgpu::BufferDescriptor { label: None, size: bytes, usage: wgpu::BufferUsages::STORAGE | wgpu::BufferUsages::COPY_SRC | wgpu::BufferUsages::COPY_DST, mapped_at_creation: false };
    let src = device.create_buffer(&desc);
    let buffer_b = device.create_buffer(&desc);
    let staging = device.create_buffer(&desc);
    let onlyinverse = Onlyinverse::new(&device, &queue, &src, &buffer_b, n as u32);
    let normalize = Normalize::new(&device, &queue, &src, &buffer_b, n as u32);
    queue.write_buffer(&src, 0, bytemuck::cast_slice(&data));
    let mut encoder = device.create_command_encoder(&Default::default());
    let _ = onlyinverse.proc(&mut encoder);
    let output = normalize.proc(&mut encoder);
    encoder.copy_buffer_to_buffer(output, 0, &staging, 0, bytes);
    queue.submit(Some(encoder.finish()));
    device.poll(wgpu::Maintain::wait());
    let mut ans = vec![Complex::zero(); count];
    staging.read_to(bytemuck::cast_slice_mut(&mut ans));
    let mut worst = 0f32;
    for (i, v) in ans.iter().enumerate() {
        let (er, ei) = if i % n == 0 { (data[0].real, data[0].imag) } else { (0.0, 0.0) };
        worst = worst.max((v.real - er).abs()).max((v.imag - ei).abs());
    }
    assert!(worst < 1e-5, "max error {worst}");
    println!("max error {worst}");
}
