//! processor.rs -- UNVERIFIED SOURCE (never compiled here).  The four plan structs of the reference
//! (`src/processor.rs`: `Forward` :7-159, `Inverse` :231-341, `Normalize` :409-505, `Onlyinverse` :566-670) with
//! their exact public signatures, each owning one `fwa_plan` of `include/fft_wgpu_amd.h`:
//!
//!   X::new(&'a wgpu::Device, &'a wgpu::Queue, &'a wgpu::Buffer [, &'a wgpu::Buffer], fft_len: u32) -> Self
//!   X::proc(&self, encoder: &mut wgpu::CommandEncoder) -> &wgpu::Buffer
//!
//! `proc` enqueues the transform on the encoder's stream and returns the buffer that will hold the natural-order
//! result, by the reference's rule (`processor.rs:153-157,335-339,433-439,664-668`): `src` when log2(fft_len) is even,
//! the second buffer otherwise.  Non-power-of-two lengths and sizes that are not a multiple of `8 * fft_len` panic
//! in `new` (the reference silently mis-computes).
use crate::ffi::*;
use crate::wgpu_helper as wgpu;
use std::ptr;

fn make_plan(device: &wgpu::Device, kind: i32, fft_len: u32, a: &wgpu::Buffer, b: Option<&wgpu::Buffer>) -> *mut fwa_plan {
    let mut p: *mut fwa_plan = ptr::null_mut();
    let b_h = b.map(|x| x.h.get()).unwrap_or(ptr::null_mut());
    let st = unsafe { fwa_plan_create(device.ctx, kind, fft_len, a.h.get(), b_h, &mut p) };
    if st != FWA_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(fwa_last_error_string(device.ctx)) }.to_string_lossy().into_owned();
        panic!("fwa_plan_create: status {st}: {msg}");
    }
    p
}

/// Enqueue and return the handle of the result buffer.
fn exec(plan: *mut fwa_plan, encoder: &mut wgpu::CommandEncoder) -> *mut fwa_buf {
    let mut res: *mut fwa_buf = ptr::null_mut();
    let st = unsafe { fwa_plan_exec(plan, encoder.s, &mut res) };
    if st != FWA_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(fwa_last_error_string(encoder.ctx)) }.to_string_lossy().into_owned();
        panic!("fwa_plan_exec: status {st}: {msg}");
    }
    res
}

/// A non-owning view of a plan-owned buffer (the reference's `buffer_b: wgpu::Buffer` field).  Its handle sits in a
/// `Cell` (`wgpu::Buffer::h`): `proc(&self)` fills it in on first use without writing through a shared reference.
fn borrowed(ctx: *mut fwa_ctx, h: *mut fwa_buf) -> wgpu::Buffer {
    wgpu::Buffer::view_of(ctx, h)
}

pub struct Forward<'a> {
    device: &'a wgpu::Device,
    #[allow(dead_code)]
    queue: &'a wgpu::Queue, // unused, as in the reference (processor.rs:9)
    plan: *mut fwa_plan,
    pub buffer_a: &'a wgpu::Buffer,
    buffer_b: wgpu::Buffer, // plan-owned ping-pong partner (processor.rs:34-41); a view of the plan's buffer
    pub fft_len: u32,
    pub data_len: u32,
}

impl<'a> Forward<'a> {
    pub fn new(device: &'a wgpu::Device, queue: &'a wgpu::Queue, src: &'a wgpu::Buffer, fft_len: u32) -> Self {
        let plan = make_plan(device, FWA_FORWARD, fft_len, src, None);
        Self {
            device,
            queue,
            plan,
            buffer_a: src,
            buffer_b: borrowed(device.ctx, ptr::null_mut()),
            fft_len,
            data_len: src.size() as u32, // the reference's truncating field (processor.rs:31); sizes here are 64-bit
        }
    }

    pub fn proc(&self, encoder: &mut wgpu::CommandEncoder) -> &wgpu::Buffer {
        let res = exec(self.plan, encoder);
        if res == self.buffer_a.h.get() {
            self.buffer_a
        } else {
            // odd log2(fft_len): the plan-owned partner.  The handle is stable for the plan's lifetime; the view's
            // `Cell` takes it on first use (interior mutability: no write through `&self`).
            self.buffer_b.h.set(res);
            let _ = self.device;
            &self.buffer_b
        }
    }
}
impl<'a> Drop for Forward<'a> {
    fn drop(&mut self) {
        unsafe { fwa_plan_destroy(self.plan) };
    }
}

pub struct Inverse<'a> {
    #[allow(dead_code)]
    device: &'a wgpu::Device,
    #[allow(dead_code)]
    queue: &'a wgpu::Queue,
    plan: *mut fwa_plan,
    buffer_a: &'a wgpu::Buffer,
    buffer_b: wgpu::Buffer,
    pub fft_len: u32,
}

impl<'a> Inverse<'a> {
    pub fn new(device: &'a wgpu::Device, queue: &'a wgpu::Queue, src: &'a wgpu::Buffer, fft_len: u32) -> Self {
        let plan = make_plan(device, FWA_INVERSE_SCALED, fft_len, src, None);
        Self { device, queue, plan, buffer_a: src, buffer_b: borrowed(device.ctx, ptr::null_mut()), fft_len }
    }

    pub fn proc(&self, encoder: &mut wgpu::CommandEncoder) -> &wgpu::Buffer {
        let res = exec(self.plan, encoder);
        if res == self.buffer_a.h.get() {
            self.buffer_a
        } else {
            self.buffer_b.h.set(res);
            &self.buffer_b
        }
    }
}
impl<'a> Drop for Inverse<'a> {
    fn drop(&mut self) {
        unsafe { fwa_plan_destroy(self.plan) };
    }
}

pub struct Normalize<'a> {
    #[allow(dead_code)]
    device: &'a wgpu::Device,
    #[allow(dead_code)]
    queue: &'a wgpu::Queue,
    plan: *mut fwa_plan,
    buffer_a: &'a wgpu::Buffer,
    buffer_b: &'a wgpu::Buffer,
    pub fft_len: u32,
}

impl<'a> Normalize<'a> {
    pub fn new(
        device: &'a wgpu::Device,
        queue: &'a wgpu::Queue,
        buffer1: &'a wgpu::Buffer,
        buffer2: &'a wgpu::Buffer,
        fft_len: u32,
    ) -> Self {
        let plan = make_plan(device, FWA_NORMALIZE, fft_len, buffer1, Some(buffer2));
        Self { device, queue, plan, buffer_a: buffer1, buffer_b: buffer2, fft_len }
    }

    /// b[i] = a[i] / fft_len with (a, b) = (buffer1, buffer2) if log2(fft_len) is even, else swapped
    /// (processor.rs:433-439); returns b.
    pub fn proc(&self, encoder: &mut wgpu::CommandEncoder) -> &wgpu::Buffer {
        let res = exec(self.plan, encoder);
        if res == self.buffer_a.h.get() { self.buffer_a } else { self.buffer_b }
    }
}
impl<'a> Drop for Normalize<'a> {
    fn drop(&mut self) {
        unsafe { fwa_plan_destroy(self.plan) };
    }
}

pub struct Onlyinverse<'a> {
    #[allow(dead_code)]
    device: &'a wgpu::Device,
    #[allow(dead_code)]
    queue: &'a wgpu::Queue,
    plan: *mut fwa_plan,
    buffer_a: &'a wgpu::Buffer,
    buffer_b: &'a wgpu::Buffer,
    pub fft_len: u32,
}

impl<'a> Onlyinverse<'a> {
    pub fn new(
        device: &'a wgpu::Device,
        queue: &'a wgpu::Queue,
        src: &'a wgpu::Buffer,
        src2: &'a wgpu::Buffer,
        fft_len: u32,
    ) -> Self {
        let plan = make_plan(device, FWA_INVERSE_UNSCALED, fft_len, src, Some(src2));
        Self { device, queue, plan, buffer_a: src, buffer_b: src2, fft_len }
    }

    pub fn proc(&self, encoder: &mut wgpu::CommandEncoder) -> &wgpu::Buffer {
        let res = exec(self.plan, encoder);
        if res == self.buffer_a.h.get() { self.buffer_a } else { self.buffer_b }
    }
}
impl<'a> Drop for Onlyinverse<'a> {
    fn drop(&mut self) {
        unsafe { fwa_plan_destroy(self.plan) };
    }
}
