//! wgpu_helper -- UNVERIFIED SOURCE (never compiled here).  The reference declares this module (`src/lib.rs:8`) and
//! leaves it empty; here it holds the GPU layer: the `wgpu` objects the plans (`src/processor.rs`) and the examples
//! (`src/examples/basic.rs:6-122`) use, each a thin owner of one C-ABI handle of `include/fft_wgpu_amd.h`.
//! Semantics follow the reference's use of wgpu: a `CommandEncoder` records work in order (one HIP stream),
//! `Queue::submit` is the ordering point, `Device::poll(Maintain::Wait)` waits for everything submitted.
use crate::ffi::*;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::ptr;

fn check(ctx: *const fwa_ctx, st: i32, what: &str) {
    if st != FWA_OK {
        // the reference unwraps / panics on every failure (examples/basic.rs:14,30,106)
        let msg = unsafe { CStr::from_ptr(fwa_last_error_string(ctx)) }.to_string_lossy().into_owned();
        panic!("{what}: status {st}: {msg}");
    }
}

/// `wgpu::Device` (+ Instance + Adapter): one context per GPU ordinal.
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
    /// wgpu 23/24 spelling used by the reference (`src/lib.rs:226`)
    pub fn wait() -> Self {
        Maintain::Wait
    }
}

/// `wgpu::BufferDescriptor` as far as the reference fills it (`examples/basic.rs:50-64`): usage flags are accepted
/// and ignored (every buffer is device memory usable as STORAGE | COPY_SRC | COPY_DST).
pub struct BufferDescriptor<'a> {
    pub label: Option<&'a str>,
    pub size: u64,
    pub usage: u32,
    pub mapped_at_creation: bool,
}

impl Device {
    /// `None` when no gfx950 device is usable (reference `prepare_gpu` -> `None`, `src/lib.rs:43,59`).
    pub fn open(ordinal: i32) -> Option<Device> {
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
        Buffer { ctx: self.ctx, h, owned: true }
    }
    /// `create_command_encoder(&Default::default())` (`examples/basic.rs:76`)
    pub fn create_command_encoder(&self, _desc: &CommandEncoderDescriptor) -> CommandEncoder {
        let mut s: *mut fwa_stream = ptr::null_mut();
        check(self.ctx, unsafe { fwa_stream_create(self.ctx, &mut s) }, "fwa_stream_create");
        CommandEncoder { ctx: self.ctx, s }
    }
    /// `device.poll(wgpu::Maintain::wait())` (`examples/basic.rs:106`): all submitted work has completed on return.
    pub fn poll(&self, _maintain: Maintain) {
        check(self.ctx, unsafe { fwa_ctx_synchronize(self.ctx) }, "fwa_ctx_synchronize");
    }
}
impl Drop for Device {
    fn drop(&mut self) {
        if self.owner {
            unsafe { fwa_ctx_destroy(self.ctx) };
        }
    }
}

#[derive(Default)]
pub struct CommandEncoderDescriptor;

/// `wgpu::Buffer`
pub struct Buffer {
    pub(crate) ctx: *mut fwa_ctx,
    pub(crate) h: *mut fwa_buf,
    pub(crate) owned: bool,
}
impl Buffer {
    /// bytes (`src.size()`, `processor.rs:30`)
    pub fn size(&self) -> u64 {
        unsafe { fwa_buf_size(self.h) }
    }
    /// Blocking read-back: `slice(..).map_async` + `device.poll(wait)` + `get_mapped_range` + `unmap`
    /// (`examples/basic.rs:105-122`) in one call.
    pub fn read_to(&self, host: &mut [u8]) {
        check(self.ctx, unsafe { fwa_ctx_synchronize(self.ctx) }, "fwa_ctx_synchronize");
        let st = unsafe { fwa_buf_download(host.as_mut_ptr() as *mut c_void, self.h, 0, host.len() as u64, ptr::null_mut()) };
        check(self.ctx, st, "fwa_buf_download");
    }
}
impl Drop for Buffer {
    fn drop(&mut self) {
        if self.owned {
            unsafe { fwa_buf_free(self.h) };
        }
    }
}

impl Queue {
    /// `queue.write_buffer(&src, 0, bytemuck::cast_slice(&data))` (`examples/basic.rs:73`)
    pub fn write_buffer(&self, buffer: &Buffer, offset: u64, data: &[u8]) {
        let st = unsafe { fwa_buf_upload(buffer.h, offset, data.as_ptr() as *const c_void, data.len() as u64, ptr::null_mut()) };
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
        let st = unsafe { fwa_buf_copy(dst.h, dst_offset, src.h, src_offset, size, self.s) };
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
