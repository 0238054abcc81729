// ffi.rs -- the `extern "C"` block over include/fft_wgpu_amd.h (ABI version 4).  UNVERIFIED: never compiled here (no
// rustc in the image).  Generated from the header's prototypes; tests/test_abi.py checks that this list and the header
// declare the same symbols.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_void};

#[repr(C)] pub struct fwa_ctx { _private: [u8; 0] }
#[repr(C)] pub struct fwa_stream { _private: [u8; 0] }
#[repr(C)] pub struct fwa_buf { _private: [u8; 0] }
#[repr(C)] pub struct fwa_plan { _private: [u8; 0] }
#[repr(C)] pub struct fwa_event { _private: [u8; 0] }
#[repr(C)] pub struct fwa_comm { _private: [u8; 0] }

pub const FWA_COMM_ID_BYTES: usize = 128;
/// `FWA_ABI_VERSION` of the header this block was generated from; `wgpu_helper::Device::open` refuses a library that
/// reports another one.
pub const FWA_ABI_VERSION: i32 = 4;

pub const FWA_OK: i32 = 0;
pub const FWA_ERR_NO_DEVICE: i32 = 5;
pub const FWA_ERR_UNSUPPORTED: i32 = 6;
pub const FWA_FORWARD: i32 = 0;
pub const FWA_INVERSE_SCALED: i32 = 1;
pub const FWA_INVERSE_UNSCALED: i32 = 2;
pub const FWA_NORMALIZE: i32 = 3;

#[link(name = "fft_wgpu_amd")]
extern "C" {
    pub fn fwa_abi_version() -> i32;
    pub fn fwa_last_error_string(ctx: *const fwa_ctx) -> *const c_char;
    pub fn fwa_status_string(status: i32) -> *const c_char;
    pub fn fwa_device_count(count: *mut i32) -> i32;
    pub fn fwa_device_info(device_ordinal: i32, name: *mut c_char, name_cap: usize, compute_units: *mut i32, hbm_bytes: *mut u64, usable: *mut i32) -> i32;
    pub fn fwa_ctx_create(device_ordinal: i32, out: *mut *mut fwa_ctx) -> i32;
    pub fn fwa_ctx_destroy(ctx: *mut fwa_ctx) -> i32;
    pub fn fwa_ctx_synchronize(ctx: *mut fwa_ctx) -> i32;
    pub fn fwa_ctx_get_i64(ctx: *const fwa_ctx, key: *const c_char, value: *mut i64) -> i32;
    pub fn fwa_ctx_set_i64(ctx: *mut fwa_ctx, key: *const c_char, value: i64) -> i32;
    pub fn fwa_ctx_peer_access(ctx: *mut fwa_ctx, peer: *mut fwa_ctx, kind: *mut i32) -> i32;
    pub fn fwa_ctx_device_info(ctx: *const fwa_ctx, name: *mut c_char, name_cap: usize, compute_units: *mut i32, hbm_bytes: *mut u64) -> i32;
    pub fn fwa_stream_create(ctx: *mut fwa_ctx, out: *mut *mut fwa_stream) -> i32;
    pub fn fwa_stream_wrap(ctx: *mut fwa_ctx, hip_stream: *mut c_void, out: *mut *mut fwa_stream) -> i32;
    pub fn fwa_stream_synchronize(stream: *mut fwa_stream) -> i32;
    pub fn fwa_stream_destroy(stream: *mut fwa_stream) -> i32;
    pub fn fwa_buf_alloc(ctx: *mut fwa_ctx, bytes: u64, out: *mut *mut fwa_buf) -> i32;
    pub fn fwa_buf_wrap(ctx: *mut fwa_ctx, device_ptr: *mut c_void, bytes: u64, out: *mut *mut fwa_buf) -> i32;
    pub fn fwa_buf_free(buf: *mut fwa_buf) -> i32;
    pub fn fwa_buf_upload(dst: *mut fwa_buf, dst_offset: u64, host: *const c_void, bytes: u64, stream: *mut fwa_stream) -> i32;
    pub fn fwa_buf_download(host: *mut c_void, src: *const fwa_buf, src_offset: u64, bytes: u64, stream: *mut fwa_stream) -> i32;
    pub fn fwa_buf_copy(dst: *mut fwa_buf, dst_offset: u64, src: *const fwa_buf, src_offset: u64, bytes: u64, stream: *mut fwa_stream) -> i32;
    pub fn fwa_host_alloc(ctx: *mut fwa_ctx, bytes: u64, out: *mut *mut c_void) -> i32;
    pub fn fwa_host_free(ctx: *mut fwa_ctx, ptr: *mut c_void) -> i32;
    pub fn fwa_buf_download_async(host: *mut c_void, src: *const fwa_buf, src_offset: u64, bytes: u64, stream: *mut fwa_stream) -> i32;
    pub fn fwa_stream_wait_stream(stream: *mut fwa_stream, other: *mut fwa_stream) -> i32;
    pub fn fwa_buf_device_ptr(buf: *const fwa_buf) -> *mut c_void;
    pub fn fwa_buf_size(buf: *const fwa_buf) -> u64;
    pub fn fwa_plan_create(ctx: *mut fwa_ctx, kind: i32, fft_len: u32, src: *mut fwa_buf, src2_or_null: *mut fwa_buf, out: *mut *mut fwa_plan) -> i32;
    pub fn fwa_plan_exec(plan: *mut fwa_plan, stream: *mut fwa_stream, result: *mut *mut fwa_buf) -> i32;
    pub fn fwa_plan_destroy(plan: *mut fwa_plan) -> i32;
    pub fn fwa_describe_path(fft_len: u32, path: *mut i32, log2_factors: *mut u32) -> i32;
    pub fn fwa_plan_get_i64(plan: *const fwa_plan, key: *const c_char, value: *mut i64) -> i32;
    pub fn fwa_plan_set_i64(plan: *mut fwa_plan, key: *const c_char, value: i64) -> i32;
    pub fn fwa_slab(batch: u64, rank: i32, world: i32, first: *mut u64, count: *mut u64) -> i32;
    pub fn fwa_comm_pieces(batch: u64, fft_len: u32, root: i32, rank: i32, world: i32, offset: *mut u64, bytes: *mut u64, peer: *mut i32, n_pieces: *mut i32) -> i32;
    pub fn fwa_comm_unique_id(id: *mut u8) -> i32;
    pub fn fwa_comm_create(ctx: *mut fwa_ctx, id: *const u8, world: i32, rank: i32, out: *mut *mut fwa_comm) -> i32;
    pub fn fwa_comm_destroy(comm: *mut fwa_comm) -> i32;
    pub fn fwa_comm_get_i64(comm: *const fwa_comm, key: *const c_char, value: *mut i64) -> i32;
    pub fn fwa_comm_sendrecv(comm: *mut fwa_comm, send: *const fwa_buf, send_offset: u64, send_bytes: u64, send_to: i32, recv: *mut fwa_buf, recv_offset: u64, recv_bytes: u64, recv_from: i32, stream: *mut fwa_stream) -> i32;
    pub fn fwa_comm_scatter(comm: *mut fwa_comm, root: i32, full_or_null: *const fwa_buf, slab: *mut fwa_buf, fft_len: u32, batch: u64, stream: *mut fwa_stream) -> i32;
    pub fn fwa_comm_gather(comm: *mut fwa_comm, root: i32, slab: *const fwa_buf, full_or_null: *mut fwa_buf, fft_len: u32, batch: u64, stream: *mut fwa_stream) -> i32;
    pub fn fwa_event_create(ctx: *mut fwa_ctx, out: *mut *mut fwa_event) -> i32;
    pub fn fwa_event_record(ev: *mut fwa_event, stream: *mut fwa_stream) -> i32;
    pub fn fwa_event_synchronize(ev: *mut fwa_event) -> i32;
    pub fn fwa_stream_wait_event(stream: *mut fwa_stream, ev: *mut fwa_event) -> i32;
    pub fn fwa_event_elapsed_ms(start: *mut fwa_event, end: *mut fwa_event, ms: *mut f32) -> i32;
    pub fn fwa_event_destroy(ev: *mut fwa_event) -> i32;
    pub fn fwa_fill_synthetic(dst: *mut fwa_buf, seed: u64, first_transform: u64, fft_len: u32, scale: f32, stream: *mut fwa_stream) -> i32;
    pub fn fwa_calib_copy(dst: *mut fwa_buf, src: *const fwa_buf, bytes: u64, stream: *mut fwa_stream) -> i32;
}
