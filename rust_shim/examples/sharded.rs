//! sharded.rs -- UNVERIFIED SOURCE (never compiled here: no Rust toolchain).  One batch over every GPU of the node from one
//! process, written against this crate's stand-in for `wgpu` plus `fft_wgpu::sharded`: enumerate the adapters as the reference
//! does (`src/lib.rs:33-35`), open one device per adapter, give each its slab, `proc` on all of them, wait, read back.  The
//! compiled twin is `tools/example_sharded.cpp` (checked bit for bit against the unsharded transform by the GPU tests).
use fft_wgpu::sharded::{self, ShardedBatch};
use fft_wgpu::{wgpu, Complex, Forward};

fn main() {
    let (fft_len, batch) = (1u32 << 20, 64u64);
    let instance = wgpu::Instance::default();
    for adapter in instance.enumerate_adapters(wgpu::Backends::VULKAN) {
        println!("{:?}", adapter.get_info());
    }
    // one Device + Queue + slab Buffer per adapter; slab r = sharded::slab(batch, r, shards.len())
    let shards = sharded::open_shards(&instance, fft_len, batch);
    assert!(!shards.is_empty(), "no usable gfx950 device");
    let input: Vec<Complex> = (0..batch * fft_len as u64).map(|i| Complex::new((i % 7) as f32 - 3.0, (i % 5) as f32 - 2.0)).collect();
    for s in &shards {
        let lo = (s.first * fft_len as u64) as usize;
        let hi = lo + (s.count * fft_len as u64) as usize;
        s.queue.write_buffer(&s.src, 0, bytemuck::cast_slice(&input[lo..hi]));
    }
    let plans = ShardedBatch::<Forward>::new(&shards, fft_len);
    let mut encoders = sharded::encoders(&shards);
    let results = plans.proc(&mut encoders); // enqueued on every GPU; nothing communicates
    sharded::poll_all(&shards);
    let mut output = vec![Complex::zero(); input.len()];
    for (s, r) in shards.iter().zip(results.iter()) {
        let lo = (s.first * fft_len as u64) as usize;
        let hi = lo + (s.count * fft_len as u64) as usize;
        r.read_to(bytemuck::cast_slice_mut(&mut output[lo..hi])); // map_async + poll + get_mapped_range + unmap, folded
    }
    println!("{} transforms of {} points over {} GPU(s); X[0] = {:?}", batch, fft_len, shards.len(), output[0]);
}
