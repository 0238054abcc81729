//! sharded.rs -- UNVERIFIED SOURCE (never compiled here: no Rust toolchain).  Batch sharding over the GPUs of one node for
//! callers of this crate; the compiled and tested twins are `fft_wgpu::ShardedBatch` (`include/fft_wgpu.hpp`) and
//! `fft_wgpu_amd.ShardedBatch` (Python).  The reference drives one device and one queue (`src/lib.rs:29-62`) and has no
//! counterpart; what makes sharding trivial is that a transform reads only its own samples
//! (`src/kernel/fft4.wgsl:21-23`), so a batch splits into contiguous slabs of whole transforms and nothing communicates
//! during `proc`.
//!
//!   let instance = wgpu::Instance::default();
//!   let shards = sharded::open_shards(&instance, fft_len, batch);       // one Device + Queue + slab Buffer per adapter
//!   let plans = sharded::ShardedBatch::<Forward>::new(&shards, fft_len);
//!   let mut encoders = sharded::encoders(&shards);
//!   let results = plans.proc(&mut encoders);                            // enqueued on every GPU, returns at once
//!   sharded::poll_all(&shards);                                         // join
//!
//! Moving slabs: one process holds every pointer, so `encoder.copy_buffer_to_buffer(&full, first * 8 * n, &shard.src, 0, ..)`
//! works across devices (an explicit peer copy inside `fwa_buf_copy`, `FWA_ERR_UNSUPPORTED` when the devices cannot reach each
//! other).  One process PER GPU uses `Comm` (`fwa_comm_*`: RCCL grouped send / receive) instead.
use crate::ffi::*;
use crate::processor::{Forward, Inverse};
use crate::wgpu_helper as wgpu;
use std::ptr;

/// The slab rule (`fwa_slab`): rank `rank` of `world` owns transforms `[first, first + count)`.
pub fn slab(batch: u64, rank: i32, world: i32) -> (u64, u64) {
    let (mut first, mut count) = (0u64, 0u64);
    let st = unsafe { fwa_slab(batch, rank, world, &mut first, &mut count) };
    assert!(st == FWA_OK, "fwa_slab: bad rank / world");
    (first, count)
}

/// One GPU's share of a batch: its device, queue and slab buffer.
pub struct Shard {
    pub device: wgpu::Device,
    pub queue: wgpu::Queue,
    pub src: wgpu::Buffer,
    pub first: u64,
    pub count: u64,
}

/// One shard per adapter of `instance.enumerate_adapters(..)` (`src/lib.rs:33-35`), slab buffers allocated.
pub fn open_shards(instance: &wgpu::Instance, fft_len: u32, batch: u64) -> Vec<Shard> {
    let adapters = instance.enumerate_adapters(wgpu::Backends::all());
    let world = adapters.len() as i32;
    adapters
        .iter()
        .enumerate()
        .map(|(r, a)| {
            let device = wgpu::Device::open(a.ordinal()).expect("adapter was enumerated as usable");
            let queue = device.queue();
            let (first, count) = slab(batch, r as i32, world);
            let src = device.create_buffer(&wgpu::BufferDescriptor {
                label: None,
                size: count * 8 * fft_len as u64,
                usage: wgpu::BufferUsages::STORAGE | wgpu::BufferUsages::COPY_SRC | wgpu::BufferUsages::COPY_DST,
                mapped_at_creation: false,
            });
            Shard { device, queue, src, first, count }
        })
        .collect()
}

pub fn encoders(shards: &[Shard]) -> Vec<wgpu::CommandEncoder> {
    shards.iter().map(|s| s.device.create_command_encoder(&wgpu::CommandEncoderDescriptor { label: None })).collect()
}

pub fn poll_all(shards: &[Shard]) {
    for s in shards {
        s.device.poll(wgpu::Maintain::wait()).panic_on_timeout();
    }
}

/// The plan kinds that own their second buffer (`Forward`, `Inverse`): what a sharded batch is made of.
pub trait ShardPlan<'a>: Sized {
    fn new_on(shard: &'a Shard, fft_len: u32) -> Self;
    fn proc_on(&self, encoder: &mut wgpu::CommandEncoder) -> &wgpu::Buffer;
}
impl<'a> ShardPlan<'a> for Forward<'a> {
    fn new_on(s: &'a Shard, fft_len: u32) -> Self {
        Forward::new(&s.device, &s.queue, &s.src, fft_len)
    }
    fn proc_on(&self, e: &mut wgpu::CommandEncoder) -> &wgpu::Buffer {
        self.proc(e)
    }
}
impl<'a> ShardPlan<'a> for Inverse<'a> {
    fn new_on(s: &'a Shard, fft_len: u32) -> Self {
        Inverse::new(&s.device, &s.queue, &s.src, fft_len)
    }
    fn proc_on(&self, e: &mut wgpu::CommandEncoder) -> &wgpu::Buffer {
        self.proc(e)
    }
}

/// One plan per shard.
pub struct ShardedBatch<'a, P: ShardPlan<'a>> {
    pub plans: Vec<P>,
    pub shards: &'a [Shard],
}
impl<'a, P: ShardPlan<'a>> ShardedBatch<'a, P> {
    pub fn new(shards: &'a [Shard], fft_len: u32) -> Self {
        Self { plans: shards.iter().map(|s| P::new_on(s, fft_len)).collect(), shards }
    }
    /// Enqueue every shard's transform on its own encoder; the result buffers in shard order.
    pub fn proc(&self, encoders: &mut [wgpu::CommandEncoder]) -> Vec<&wgpu::Buffer> {
        self.plans.iter().zip(encoders.iter_mut()).map(|(p, e)| p.proc_on(e)).collect()
    }
}

/// `fwa_comm_*`: one rank of a slab communicator over RCCL, for one process per GPU.
pub struct Comm {
    h: *mut fwa_comm,
    pub world: i32,
    pub rank: i32,
}
impl Comm {
    /// Rank 0 makes the id and hands the bytes to the other ranks (file, socket, environment).
    pub fn unique_id() -> [u8; FWA_COMM_ID_BYTES] {
        let mut id = [0u8; FWA_COMM_ID_BYTES];
        assert!(unsafe { fwa_comm_unique_id(id.as_mut_ptr()) } == FWA_OK, "fwa_comm_unique_id");
        id
    }
    /// Collective over the `world` processes that hold `id`.
    pub fn new(device: &wgpu::Device, id: &[u8; FWA_COMM_ID_BYTES], world: i32, rank: i32) -> Self {
        let mut h: *mut fwa_comm = ptr::null_mut();
        let st = unsafe { fwa_comm_create(device.ctx, id.as_ptr(), world, rank, &mut h) };
        assert!(st == FWA_OK, "fwa_comm_create: status {st}");
        Self { h, world, rank }
    }
    /// root's `full` (ignored elsewhere) -> every rank's `slab`, ordered on `encoder`'s stream.
    pub fn scatter(&self, root: i32, full: Option<&wgpu::Buffer>, slab: &wgpu::Buffer, fft_len: u32, batch: u64, encoder: &mut wgpu::CommandEncoder) {
        let f = full.map(|b| b.h.get() as *const fwa_buf).unwrap_or(ptr::null());
        let st = unsafe { fwa_comm_scatter(self.h, root, f, slab.h.get(), fft_len, batch, encoder.s) };
        assert!(st == FWA_OK, "fwa_comm_scatter: status {st}");
    }
    pub fn gather(&self, root: i32, slab: &wgpu::Buffer, full: Option<&wgpu::Buffer>, fft_len: u32, batch: u64, encoder: &mut wgpu::CommandEncoder) {
        let f = full.map(|b| b.h.get()).unwrap_or(ptr::null_mut());
        let st = unsafe { fwa_comm_gather(self.h, root, slab.h.get(), f, fft_len, batch, encoder.s) };
        assert!(st == FWA_OK, "fwa_comm_gather: status {st}");
    }
}
impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { fwa_comm_destroy(self.h) };
    }
}
