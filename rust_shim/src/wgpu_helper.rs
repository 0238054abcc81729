//! wgpu_helper -- UNVERIFIED SOURCE (never compiled here: no Rust toolchain in the image or on the GPU box).  The
//! reference declares this module (`src/lib.rs:8`) and leaves it empty; here it holds the GPU layer: every `wgpu` item the
//! plans (`src/processor.rs`) and the three examples (`src/examples/basic.rs:6-30,50-64,68,73-122` and the same lines of
//! `basic_inverse.rs`, `basic_inverse2.rs`) name, each a thin owner of one C-ABI handle of `include/fft_wgpu_amd.h`:
//!
//!   Instance::default() -> enumerate_adapters(Backends::VULKAN) -> Vec<Adapter> (one per GPU ordinal; adapter.get_info())
//!   Instance::default() -> request_adapter(&RequestAdapterOptions { power_preference, .. }).await -> Option<Adapter>
//!   adapter.features() / limits(); adapter.request_device(&DeviceDescriptor { .. }, None).await -> Result<(Device, Queue), _>
//!   device.create_buffer(&BufferDescriptor { label, size, usage: BufferUsages::A | BufferUsages::B, mapped_at_creation })
//!   buffer.slice(..) -> BufferSlice; slice.map_async(MapMode::Read, |_| {}); device.poll(Maintain::wait()).panic_on_timeout();
//!   slice.get_mapped_range() -> BufferView (Deref<Target = [u8]>); buffer.unmap()
//!   device.create_command_encoder(&CommandEncoderDescriptor { label }); encoder.copy_buffer_to_buffer(..); encoder.finish()
//!   queue.write_buffer(&buf, offset, bytes); queue.submit(Some(command_buffer))
//!
//! Semantics follow the reference's use of wgpu: a `CommandEncoder` records work in order (one HIP stream), `Queue::submit`
//! is the ordering point, `Device::poll(Maintain::Wait)` waits for everything submitted, a mapped range is a host copy of
//! the buffer taken after that wait.  The futures are ready on first poll (there is nothing to wait for: the HIP context
//! is created synchronously), so `#[tokio::main]` callers keep their `.await`s.
use crate::ffi::*;
use std::cell::{Cell, RefCell};
use std::ffi::CStr;
use std::future::{ready, Ready};
use std::ops::{BitOr, Deref, RangeBounds};
use std::os::raw::c_void;
use std::ptr;

fn check(ctx: *const fwa_ctx, st: i32, what: &str) {
    if st != FWA_OK {
        // the reference unwraps / panics on every failure (examples/basic.rs:14,30,106)
        let msg = unsafe { CStr::from_ptr(fwa_last_error_string(ctx)) }.to_string_lossy().into_owned();
        panic!("{what}: status {st}: {msg}");
    }
}

// ---- Instance / Adapter (examples/basic.rs:6-30, src/lib.rs:29-62) -------------------------------------------------
/// `wgpu::Instance`: nothing to hold -- HIP needs no loader object.
#[derive(Default)]
pub struct Instance;

/// `wgpu::PowerPreference`
#[derive(Clone, Copy, Default, PartialEq, Eq, Debug)]
pub enum PowerPreference {
    #[default]
    None,
    LowPower,
    HighPerformance,
}

/// `wgpu::RequestAdapterOptions` as far as the reference fills it (`power_preference`, the rest defaulted).
#[derive(Default)]
pub struct RequestAdapterOptions {
    pub power_preference: PowerPreference,
    pub force_fallback_adapter: bool,
    pub compatible_surface: Option<()>,
}

/// `wgpu::Features` / `wgpu::Limits`: opaque here (the reference only passes the adapter's own values back).
#[derive(Clone, Copy, Default, Debug)]
pub struct Features;
#[derive(Clone, Copy, Default, Debug)]
pub struct Limits;

/// `wgpu::DeviceDescriptor`
#[derive(Default)]
pub struct DeviceDescriptor<'a> {
    pub label: Option<&'a str>,
    pub required_features: Features,
    pub required_limits: Limits,
}

#[derive(Debug)]
pub struct RequestDeviceError(pub String);

/// `wgpu::Backends`: accepted and ignored -- there is one backend here (HIP on gfx950).  The reference lists its adapters per
/// backend (`src/lib.rs:33-35`: `VULKAN`, `GL`, `METAL`); to keep such a listing free of duplicates only the backends an
/// MI355X host would really answer on (`VULKAN`, `PRIMARY`, `all()`) return the devices, the others an empty list.
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub struct Backends(pub u32);
impl Backends {
    pub const VULKAN: Backends = Backends(1 << 0);
    pub const GL: Backends = Backends(1 << 1);
    pub const METAL: Backends = Backends(1 << 2);
    pub const DX12: Backends = Backends(1 << 3);
    pub const BROWSER_WEBGPU: Backends = Backends(1 << 4);
    pub const PRIMARY: Backends = Backends(1 << 0 | 1 << 2 | 1 << 3 | 1 << 4);
    pub const fn all() -> Self {
        Backends(0x1f)
    }
}

/// `wgpu::AdapterInfo` as far as it can be filled from `fwa_device_info`.
#[derive(Clone, Debug)]
pub struct AdapterInfo {
    pub name: String,        // gcnArchName, e.g. "gfx950:sramecc+:xnack-"
    pub device: u32,         // the HIP device ordinal
    pub compute_units: i32,
    pub hbm_bytes: u64,
}

/// `wgpu::Adapter`: a GPU ordinal that was found usable.
#[derive(Debug)]
pub struct Adapter {
    ordinal: i32,
}

impl Instance {
    /// `instance.enumerate_adapters(Backends::VULKAN)` (`src/lib.rs:33-35`): one adapter per usable GPU ordinal, in ordinal
    /// order -- what a multi-GPU caller iterates to open one `Device` per GPU (`sharded::ShardedBatch`).
    pub fn enumerate_adapters(&self, backends: Backends) -> Vec<Adapter> {
        let mut out = Vec::new();
        if backends.0 & Backends::VULKAN.0 == 0 {
            return out;
        }
        let mut n: i32 = 0;
        if unsafe { fwa_device_count(&mut n) } != FWA_OK {
            return out;
        }
        for ordinal in 0..n {
            let mut usable: i32 = 0;
            let st = unsafe { fwa_device_info(ordinal, ptr::null_mut(), 0, ptr::null_mut(), ptr::null_mut(), &mut usable) };
            if st == FWA_OK && usable != 0 {
                out.push(Adapter { ordinal });
            }
        }
        out
    }

    /// `None` when no gfx950 device is visible (`prepare_gpu` -> `None`, `src/lib.rs:43`); otherwise the first usable one
    /// (every MI355X of a node is equally "high performance").
    pub fn request_adapter(&self, _options: &RequestAdapterOptions) -> Ready<Option<Adapter>> {
        ready(self.enumerate_adapters(Backends::all()).into_iter().next())
    }
}

impl Adapter {
    /// `adapter.get_info()`
    pub fn get_info(&self) -> AdapterInfo {
        let mut name = [0 as std::os::raw::c_char; 256];
        let (mut cus, mut mem, mut usable) = (0i32, 0u64, 0i32);
        check(ptr::null(), unsafe { fwa_device_info(self.ordinal, name.as_mut_ptr(), name.len(), &mut cus, &mut mem, &mut usable) }, "fwa_device_info");
        AdapterInfo {
            name: unsafe { CStr::from_ptr(name.as_ptr()) }.to_string_lossy().into_owned(),
            device: self.ordinal as u32,
            compute_units: cus,
            hbm_bytes: mem,
        }
    }
    /// The HIP device ordinal behind this adapter.
    pub fn ordinal(&self) -> i32 {
        self.ordinal
    }
    pub fn features(&self) -> Features {
        Features
    }
    pub fn limits(&self) -> Limits {
        Limits
    }
    /// `adapter.request_device(&desc, None).await` (`examples/basic.rs:20-30`; the second argument is wgpu 23/24's trace path)
    pub fn request_device(
        &self,
        _desc: &DeviceDescriptor,
        _trace_path: Option<&std::path::Path>,
    ) -> Ready<Result<(Device, Queue), RequestDeviceError>> {
        ready(match Device::open(self.ordinal) {
            Some(d) => {
                let q = d.queue();
                Ok((d, q))
            }
            None => Err(RequestDeviceError("no usable gfx950 device".into())),
        })
    }
}

// ---- Device / Queue ------------------------------------------------------------------------------------------------
/// `wgpu::Device`: one context per GPU ordinal.
pub struct Device {
    pub(crate) ctx: *mut fwa_ctx,
    owner: bool,
}

/// `wgpu::Queue`: shares the device's context.
pub struct Queue {
    pub(crate) ctx: *mut fwa_ctx,
}

/// `wgpu::Maintain`
pub enum Maintain {
    Wait,
    Poll,
}
impl Maintain {
    /// wgpu 23/24 spelling used by the reference (`src/lib.rs:226`, `examples/basic.rs:106`)
    pub fn wait() -> Self {
        Maintain::Wait
    }
}
/// `wgpu::MaintainResult`
pub enum MaintainResult {
    SubmissionQueueEmpty,
    Ok,
}
impl MaintainResult {
    /// `device.poll(..).panic_on_timeout()` (`examples/basic.rs:106`): a failed wait has already panicked in `poll`.
    pub fn panic_on_timeout(self) {}
}

/// `wgpu::BufferUsages`: accepted and ignored -- every buffer is device memory usable as STORAGE | COPY_SRC | COPY_DST,
/// and any buffer can be read back through `slice(..).map_async` (a staged host copy).
#[derive(Clone, Copy, PartialEq, Eq, Debug, Default)]
pub struct BufferUsages(pub u32);
impl BufferUsages {
    pub const MAP_READ: BufferUsages = BufferUsages(1 << 0);
    pub const MAP_WRITE: BufferUsages = BufferUsages(1 << 1);
    pub const COPY_SRC: BufferUsages = BufferUsages(1 << 2);
    pub const COPY_DST: BufferUsages = BufferUsages(1 << 3);
    pub const INDEX: BufferUsages = BufferUsages(1 << 4);
    pub const VERTEX: BufferUsages = BufferUsages(1 << 5);
    pub const UNIFORM: BufferUsages = BufferUsages(1 << 6);
    pub const STORAGE: BufferUsages = BufferUsages(1 << 7);
    pub const fn empty() -> Self {
        BufferUsages(0)
    }
    pub const fn contains(self, other: BufferUsages) -> bool {
        self.0 & other.0 == other.0
    }
}
impl BitOr for BufferUsages {
    type Output = BufferUsages;
    fn bitor(self, rhs: BufferUsages) -> BufferUsages {
        BufferUsages(self.0 | rhs.0)
    }
}

/// `wgpu::BufferDescriptor` (`examples/basic.rs:50-64`)
pub struct BufferDescriptor<'a> {
    pub label: Option<&'a str>,
    pub size: u64,
    pub usage: BufferUsages,
    pub mapped_at_creation: bool,
}

/// `wgpu::CommandEncoderDescriptor` (`examples/basic.rs:76-77`)
#[derive(Default)]
pub struct CommandEncoderDescriptor<'a> {
    pub label: Option<&'a str>,
}

impl Device {
    /// `None` when no gfx950 device is usable (reference `prepare_gpu` -> `None`, `src/lib.rs:43,59`).
    pub fn open(ordinal: i32) -> Option<Device> {
        // A libfft_wgpu_amd.so built from another version of the header is as unusable as no GPU at all: `None`, the
        // answer the reference's `prepare_gpu` gives when nothing usable is found (`src/lib.rs:43,59`), rather than a
        // call through a changed signature later on.
        let abi = unsafe { fwa_abi_version() };
        if abi != FWA_ABI_VERSION {
            eprintln!("fft_wgpu: libfft_wgpu_amd.so reports ABI version {abi}, this crate binds version {FWA_ABI_VERSION}");
            return None;
        }
        let mut ctx: *mut fwa_ctx = ptr::null_mut();
        let st = unsafe { fwa_ctx_create(ordinal, &mut ctx) };
        if st == FWA_ERR_NO_DEVICE {
            return None;
        }
        check(ptr::null(), st, "fwa_ctx_create");
        Some(Device { ctx, owner: true })
    }
    pub fn queue(&self) -> Queue {
        Queue { ctx: self.ctx }
    }
    pub fn create_buffer(&self, desc: &BufferDescriptor) -> Buffer {
        let mut h: *mut fwa_buf = ptr::null_mut();
        check(self.ctx, unsafe { fwa_buf_alloc(self.ctx, desc.size, &mut h) }, "fwa_buf_alloc");
        Buffer { ctx: self.ctx, h: Cell::new(h), owned: true, map_requested: Cell::new(false), mapped: RefCell::new(None) }
    }
    /// `device.create_command_encoder(&wgpu::CommandEncoderDescriptor { label: None })` (`examples/basic.rs:76-77`)
    pub fn create_command_encoder(&self, _desc: &CommandEncoderDescriptor) -> CommandEncoder {
        let mut s: *mut fwa_stream = ptr::null_mut();
        check(self.ctx, unsafe { fwa_stream_create(self.ctx, &mut s) }, "fwa_stream_create");
        CommandEncoder { ctx: self.ctx, s }
    }
    /// `device.poll(wgpu::Maintain::wait())` (`examples/basic.rs:106`): all submitted work has completed on return.
    pub fn poll(&self, _maintain: Maintain) -> MaintainResult {
        check(self.ctx, unsafe { fwa_ctx_synchronize(self.ctx) }, "fwa_ctx_synchronize");
        MaintainResult::SubmissionQueueEmpty
    }
}
impl Drop for Device {
    fn drop(&mut self) {
        if self.owner {
            unsafe { fwa_ctx_destroy(self.ctx) };
        }
    }
}

// ---- Buffer / BufferSlice / BufferView -----------------------------------------------------------------------------
/// `wgpu::Buffer`.  The handle sits in a `Cell`: the plans resolve their plan-owned second buffer on the first `proc(&self)`
/// (`processor.rs`), which must not write through a shared reference without interior mutability.
pub struct Buffer {
    pub(crate) ctx: *mut fwa_ctx,
    pub(crate) h: Cell<*mut fwa_buf>,
    pub(crate) owned: bool,
    map_requested: Cell<bool>,
    mapped: RefCell<Option<Vec<u8>>>,
}
/// `wgpu::MapMode`
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum MapMode {
    Read,
    Write,
}
#[derive(Debug)]
pub struct BufferAsyncError;

/// `wgpu::BufferSlice`: the reference only ever takes the whole buffer (`staging_buffer.slice(..)`, `examples/basic.rs:68`).
pub struct BufferSlice<'a> {
    buffer: &'a Buffer,
    offset: u64,
    size: u64,
}
/// `wgpu::BufferView`: a host copy of the mapped range (`bytemuck::cast_slice(&data1)`, `examples/basic.rs:107-113`).
pub struct BufferView {
    data: Vec<u8>,
}
impl Deref for BufferView {
    type Target = [u8];
    fn deref(&self) -> &[u8] {
        &self.data
    }
}
impl AsRef<[u8]> for BufferView {
    fn as_ref(&self) -> &[u8] {
        &self.data
    }
}

impl Buffer {
    pub(crate) fn view_of(ctx: *mut fwa_ctx, h: *mut fwa_buf) -> Buffer {
        Buffer { ctx, h: Cell::new(h), owned: false, map_requested: Cell::new(false), mapped: RefCell::new(None) }
    }
    /// bytes (`src.size()`, `processor.rs:30`)
    pub fn size(&self) -> u64 {
        unsafe { fwa_buf_size(self.h.get()) }
    }
    /// `buffer.slice(..)` (`examples/basic.rs:68`)
    pub fn slice<R: RangeBounds<u64>>(&self, range: R) -> BufferSlice<'_> {
        use std::ops::Bound::*;
        let start = match range.start_bound() { Included(&a) => a, Excluded(&a) => a + 1, Unbounded => 0 };
        let end = match range.end_bound() { Included(&b) => b + 1, Excluded(&b) => b, Unbounded => self.size() };
        BufferSlice { buffer: self, offset: start, size: end - start }
    }
    /// `staging_buffer.unmap()` (`examples/basic.rs:122`): drops the host copy.
    pub fn unmap(&self) {
        self.map_requested.set(false);
        *self.mapped.borrow_mut() = None;
    }
    /// Blocking read-back in one call: `slice(..).map_async` + `device.poll(wait)` + `get_mapped_range` + `unmap`.
    pub fn read_to(&self, host: &mut [u8]) {
        check(self.ctx, unsafe { fwa_ctx_synchronize(self.ctx) }, "fwa_ctx_synchronize");
        let st = unsafe { fwa_buf_download(host.as_mut_ptr() as *mut c_void, self.h.get(), 0, host.len() as u64, ptr::null_mut()) };
        check(self.ctx, st, "fwa_buf_download");
    }
}
impl<'a> BufferSlice<'a> {
    /// `buffer_slice.map_async(wgpu::MapMode::Read, move |_| {})` (`examples/basic.rs:105`): the request is noted and the
    /// callback runs at once with `Ok(())` -- the bytes are fetched by `get_mapped_range` after the caller's `poll(wait)`.
    pub fn map_async(&self, _mode: MapMode, callback: impl FnOnce(Result<(), BufferAsyncError>) + Send + 'static) {
        self.buffer.map_requested.set(true);
        callback(Ok(()));
    }
    /// `buffer_slice.get_mapped_range()` (`examples/basic.rs:107`): waits for all submitted work (what the caller's
    /// `poll(wait)` already did) and copies the range to the host.
    pub fn get_mapped_range(&self) -> BufferView {
        assert!(self.buffer.map_requested.get(), "get_mapped_range() without map_async()");
        let mut data = vec![0u8; self.size as usize];
        check(self.buffer.ctx, unsafe { fwa_ctx_synchronize(self.buffer.ctx) }, "fwa_ctx_synchronize");
        let st = unsafe {
            fwa_buf_download(data.as_mut_ptr() as *mut c_void, self.buffer.h.get(), self.offset, self.size, ptr::null_mut())
        };
        check(self.buffer.ctx, st, "fwa_buf_download");
        BufferView { data }
    }
}
impl Drop for Buffer {
    fn drop(&mut self) {
        if self.owned {
            unsafe { fwa_buf_free(self.h.get()) };
        }
    }
}

impl Queue {
    /// `queue.write_buffer(&src, 0, bytemuck::cast_slice(&data))` (`examples/basic.rs:73`)
    pub fn write_buffer(&self, buffer: &Buffer, offset: u64, data: &[u8]) {
        let st = unsafe { fwa_buf_upload(buffer.h.get(), offset, data.as_ptr() as *const c_void, data.len() as u64, ptr::null_mut()) };
        check(self.ctx, st, "fwa_buf_upload");
        // the null stream is synchronous with respect to later encoder streams only after this wait
        check(self.ctx, unsafe { fwa_ctx_synchronize(self.ctx) }, "fwa_ctx_synchronize");
    }
    /// `queue.submit(Some(encoder.finish()))` (`examples/basic.rs:92`): work was enqueued as it was recorded; the
    /// command buffer is released once its stream has drained (`Device::poll`).
    pub fn submit<I: IntoIterator<Item = CommandBuffer>>(&self, buffers: I) {
        for cb in buffers {
            drop(cb);
        }
    }
}

/// `wgpu::CommandEncoder`: an in-order HIP stream.
pub struct CommandEncoder {
    pub(crate) ctx: *mut fwa_ctx,
    pub(crate) s: *mut fwa_stream,
}
/// `wgpu::CommandBuffer`
pub struct CommandBuffer {
    ctx: *mut fwa_ctx,
    s: *mut fwa_stream,
}
impl CommandEncoder {
    /// `encoder.copy_buffer_to_buffer(output, 0, &staging, 0, size)` (`examples/basic.rs:84-90`)
    pub fn copy_buffer_to_buffer(&mut self, src: &Buffer, src_offset: u64, dst: &Buffer, dst_offset: u64, size: u64) {
        let st = unsafe { fwa_buf_copy(dst.h.get(), dst_offset, src.h.get(), src_offset, size, self.s) };
        check(self.ctx, st, "fwa_buf_copy");
    }
    pub fn finish(self) -> CommandBuffer {
        let cb = CommandBuffer { ctx: self.ctx, s: self.s };
        std::mem::forget(self);
        cb
    }
}
impl Drop for CommandEncoder {
    fn drop(&mut self) {
        unsafe { fwa_stream_destroy(self.s) };
    }
}
impl Drop for CommandBuffer {
    fn drop(&mut self) {
        // stream destruction waits for nothing; the work already enqueued still completes (HIP semantics)
        let _ = self.ctx;
        unsafe { fwa_stream_destroy(self.s) };
    }
}
