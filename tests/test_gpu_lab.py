"""GPU tests of the LABORATORY build (fft_wgpu_amd/libfft_wgpu_amd_lab.so, `make -C fft_wgpu_amd/csrc lab`): the kernel
families that measured slower than the shipped ones and were moved out of the product library
-- the persistent 2^20 ring (path 5: the number DESIGN.md section 4 cites), the direct 16-point kernels with the
wavefront-shuffle exchange BASELINE.json's north_star names (small_reg = 3 / 2) -- and the two test knobs ring_rotate and
inject_launch_failure.  They stay correct and, where they share arithmetic with the shipped kernels, bit-identical to them;
the product library rejects their keys.  (Round 6 removed L2 teams, the 32-column 2^20 tile, the LDS radix-2 and n < 16
kernels, whose only loaders were the tests of this file: profiles/round6/lab_pruned_families.patch.)
"""
import numpy as np
import pytest

from conftest import REL_TOL  # noqa: F401
from test_gpu_parity import _check, _fixture_sizes_body, _run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import fft_wgpu_amd as fw
    got = fw.prepare_gpu(0, lab=True)
    assert got is not None, "no MI355X visible: the HIP path cannot run (there is no CPU fallback)"
    dev, queue = got
    assert dev.lab
    return fw, dev, queue


@pytest.mark.parametrize("small_reg", [2, 3])
def test_fixture_sizes_laboratory_kernels(gpu, oracle, k4, small_reg):
    """the 16-point kernels at 16 .. 4096 (small_reg = 3; the shipped kernels below and above), + the wavefront-shuffle
    exchange at n = 32 / 64 / 128 (2): every fixture size, forward / unscaled inverse / inverse."""
    _fixture_sizes_body(gpu, oracle, k4, None, small_reg)


def test_small_reg_accepts_only_its_three_values(gpu):
    fw, dev, queue = gpu
    plan = fw.Forward(dev, queue, dev.create_buffer(8 * 64 * 4), 64)
    for bad in (0, 4, -1):
        with pytest.raises(fw.FwaError) as e:
            plan.set("small_reg", bad)
        assert e.value.status == 1
    for ok in (3, 2, 1):
        plan.set("small_reg", ok)
        assert plan.get("small_reg") == ok


def test_n1m_ring_rotate_is_bit_identical(gpu, oracle):
    """Laboratory knob "ring_rotate" (the groups walk through a ring R times larger: same launches, larger cache footprint --
    the measurement behind DESIGN.md 4.2 item 7): same results bit for bit, scratch grows R-fold."""
    fw, dev, queue = gpu
    n, batch = 1 << 20, 40
    x = oracle.gen_input(n, batch, first_transform=9)
    ref, _, plan0 = _run(fw, dev, queue, "Forward", x, n, group=4, streams=2)
    mx, _ = oracle.compare(ref[:n], oracle.dft_f64(x[:n], n, -1))
    assert mx <= REL_TOL
    for rot in (2, 5):
        y, _, plan = _run(fw, dev, queue, "Forward", x, n, group=4, streams=2, ring_rotate=rot)
        assert plan.get("ring_rotate") == rot and plan.get("scratch_bytes") == rot * plan0.get("scratch_bytes")
        assert np.array_equal(y.view(np.uint64), ref.view(np.uint64))


@pytest.mark.parametrize("batch,depth,slots,wgs", [(4, 8, 12, 512), (5, 1, 2, 512), (17, 4, 5, 512), (40, 8, 12, 300),
                                                    (23, 2, 7, 64), (64, 8, 9, 512), (9, 16, 20, 700)])
def test_n1m_persistent_ring_pipeline(gpu, oracle, batch, depth, slots, wgs):
    """path 5: the two passes as ONE persistent launch (ticket queue, in-launch hand-offs through a small ring).
    Same arithmetic as the per-group launches: bit-identical results, and no bounded spin may have timed out."""
    fw, dev, queue = gpu
    n = 1 << 20
    x = oracle.gen_input(n, batch, first_transform=batch)
    ref, _, _ = _run(fw, dev, queue, "Forward", x, n, path=1, tile_w=16)
    _check(oracle, ref, oracle.dft_f64(x, n, -1), n)
    for rep in range(2):
        y, which, plan = _run(fw, dev, queue, "Forward", x, n, path=5, depth=depth, ring_slots=slots, wgs=wgs)
        assert which == 0 and plan.get("path") == 5 and plan.get("launches_per_exec") == 1
        assert plan.get("device_error") == 0
        bad = np.flatnonzero(y.view(np.uint64) != ref.view(np.uint64))
        assert bad.size == 0, (batch, depth, slots, wgs, rep, bad.size, bad[:8])
    z, _, plan = _run(fw, dev, queue, "Inverse", ref, n, path=5, depth=depth, ring_slots=slots, wgs=wgs)
    assert plan.get("device_error") == 0
    _check(oracle, z, x.astype(np.complex128), n)


def test_n1m_persistent_ring_large_batch_bit_identical(gpu, oracle):
    """768 MiB batch (far beyond L2 and Infinity Cache), ring slots reused 8 times, back-to-back execs: a stale cache
    line anywhere in the in-launch hand-off shows up as a mismatching 128-byte line."""
    fw, dev, queue = gpu
    n, batch = 1 << 20, 96
    x = oracle.gen_input(n, batch, first_transform=7)
    ref, _, _ = _run(fw, dev, queue, "Forward", x, n, path=1)
    for kw in (dict(), dict(depth=4, ring_slots=6), dict(depth=8, ring_slots=12, wgs=256), dict(depth=2, ring_slots=3)):
        for rep in range(2):
            y, _, plan = _run(fw, dev, queue, "Forward", x, n, path=5, **kw)
            assert plan.get("device_error") == 0
            bad = np.flatnonzero(y.view(np.uint64) != ref.view(np.uint64))
            assert bad.size == 0, (kw, rep, bad.size, bad[:8])


@pytest.mark.parametrize("n,batch,group,streams,fail_at", [(1 << 20, 40, 4, 2, 7), (1 << 20, 40, 4, 2, 0), (1 << 18, 64, 8, 2, 5), (1 << 20, 12, 4, 1, 2)])
def test_failed_launch_mid_exec_still_joins_the_chains(gpu, oracle, n, batch, group, streams, fail_at):
    """when a launch fails in the middle of an exec (forced here with the laboratory knob
    "inject_launch_failure": group `fail_at` is not launched, once), fwa_plan_exec returns FWA_ERR_LAUNCH -- and the groups
    enqueued before the failure, which run on the context's chain streams, are still joined to the caller's stream: a copy of
    the result buffer enqueued on that stream right after the failed call sees every one of them finished (bit-identical to
    a clean run), the groups from the failed one on untouched.  The plan works again afterwards (the failure is one-shot)."""
    fw, dev, queue = gpu
    x = oracle.gen_input(n, batch, first_transform=21)
    ref, _, _ = _run(fw, dev, queue, "Forward", x, n, group=group, streams=streams)
    src = dev.create_buffer(x.nbytes)
    snap = dev.create_buffer(x.nbytes)
    enc = dev.create_command_encoder()
    plan = fw.Forward(dev, queue, src, n)
    for k, v in (("group", group), ("streams", streams), ("inject_launch_failure", fail_at)):
        plan.set(k, v)
    queue.write_buffer(src, 0, x, encoder=enc)
    with pytest.raises(fw.FwaError) as e:
        plan.proc(enc)
    assert e.value.status == 4                                                   # FWA_ERR_LAUNCH
    enc.copy_buffer_to_buffer(src, 0, snap, 0, x.nbytes)                         # ordered behind the chains only through the join
    y = snap.map_read(stream=enc)
    done = fail_at * group * n                                                   # samples of the groups enqueued before the failure
    assert np.array_equal(y[:done].view(np.uint64), ref[:done].view(np.uint64))
    assert np.array_equal(y[done:].view(np.uint64), x[done:].view(np.uint64))   # in place: never launched = still the input
    queue.write_buffer(src, 0, x, encoder=enc)
    z = plan.proc(enc).map_read(stream=enc)
    assert np.array_equal(z.view(np.uint64), ref.view(np.uint64))
