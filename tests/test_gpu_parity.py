"""GPU parity tests (run on a real MI355X: `pytest -m gpu`).

Every transform goes through the C ABI (include/fft_wgpu_amd.h) via the thin
ctypes mirror; the CPU oracle (oracle/) is only the checker.

Tolerance (BASELINE.json north_star): <= 1e-5 relative fp32, measured per
transform as max_k|y-r| / max_k|r| and rel-L2 against the fp64 DFT (SURVEY.md
8(c)).  Integer/layout facts (result-buffer rule, generator) are bit-exact.
"""
import numpy as np
import pytest

from conftest import REF_ABS_TOL, REL_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import fft_wgpu_amd as fw
    got = fw.prepare_gpu(0)
    assert got is not None, "no MI355X visible: the HIP path cannot run (there is no CPU fallback)"
    dev, queue = got
    return fw, dev, queue


def _upload(fw, dev, queue, x):
    buf = dev.create_buffer(x.nbytes)
    queue.write_buffer(buf, 0, x)
    return buf


def _check(oracle, y, r, n, tol=REL_TOL):
    worst = (0.0, 0.0)
    for t in range(y.size // n):
        mx, l2 = oracle.compare(y[t * n:(t + 1) * n], r[t * n:(t + 1) * n])
        assert mx <= tol and l2 <= tol, (n, t, mx, l2)
        worst = (max(worst[0], mx), max(worst[1], l2))
    return worst


def _run(fw, dev, queue, kind, x, n, path=None, group=None, streams=None, depth=None, wgs=None, policy=None, mix=None,
         small_reg=None):
    """reference call sequence (examples/basic.rs:73-122): write_buffer -> proc -> read back"""
    src = _upload(fw, dev, queue, x)
    src2 = dev.create_buffer(x.nbytes) if kind in ("Onlyinverse",) else None
    plan = {"Forward": lambda: fw.Forward(dev, queue, src, n),
            "Inverse": lambda: fw.Inverse(dev, queue, src, n),
            "Onlyinverse": lambda: fw.Onlyinverse(dev, queue, src, src2, n)}[kind]()
    if path is not None:
        plan.set("path", path)
    if group is not None:
        plan.set("group", group)
    if streams is not None:
        plan.set("streams", streams)
    if depth is not None:
        plan.set("depth", depth)
    if wgs is not None:
        plan.set("wgs", wgs)
    if policy is not None:
        plan.set("policy", policy)
    if mix is not None:
        plan.set("mix", mix)
    if small_reg is not None:
        plan.set("small_reg", small_reg)
    enc = dev.create_command_encoder()
    out = plan.proc(enc)
    queue.submit(enc.finish())
    y = out.map_read(stream=enc)
    which = 0 if (out is src or out.device_ptr == src.device_ptr) else 1
    assert plan.get("device_error") == 0
    return y, which, plan


# ---- K1: the reference's own known answers, through the reference's call sequence ----
def test_reference_known_answers(gpu, known_answers):
    fw, dev, queue = gpu
    batch = 500 * 5  # examples/basic_inverse.rs:160 uses 512*500*5 samples
    for case in known_answers["cases"]:
        n = case["n"]
        c = np.complex64(complex(*case["c"]))
        x = np.full(n * batch, c, dtype=np.complex64)
        expect = np.zeros(n * batch, dtype=np.complex64)
        if case["plan"] == "Forward":
            y, which, _ = _run(fw, dev, queue, "Forward", x, n)
            expect[::n] = c * n
        elif case["plan"] == "Inverse":
            y, which, _ = _run(fw, dev, queue, "Inverse", x, n)
            expect[::n] = c
        else:
            # examples/basic_inverse2.rs:76-92: Onlyinverse then Normalize in one encoder
            src = _upload(fw, dev, queue, x)
            src2 = dev.create_buffer(x.nbytes)
            oi = fw.Onlyinverse(dev, queue, src, src2, n)
            nm = fw.Normalize(dev, queue, src, src2, n)
            enc = dev.create_command_encoder()
            out1 = oi.proc(enc)
            out2 = nm.proc(enc)
            queue.submit(enc.finish())
            y = out2.map_read(stream=enc)
            which = 0 if out1 is src else 1
            # processor.rs:433-439: normalize writes the buffer the inverse did NOT end in
            assert (out2 is src2) == (out1 is src)
            expect[::n] = c
        err = max(np.abs(y.real - expect.real).max(), np.abs(y.imag - expect.imag).max())
        assert err < REF_ABS_TOL, (case, err)  # examples/basic_inverse.rs:238-253
        assert which == int(np.log2(n)) % 2    # processor.rs:153-157


# ---- K4: numpy float64 fixtures, every power of two 2..1024: register radix-16 kernel (default), LDS radix-2
# kernel (small_reg=0), the wavefront-shuffle exchange at n = 32/64/128 (small_reg=2) and the literal
# one-launch-per-stage recurrence (path=2) ----
@pytest.mark.parametrize("path,small_reg", [(None, 1), (None, 2), (None, 0), (2, None)])
def test_fixture_sizes(gpu, oracle, k4, path, small_reg):
    fw, dev, queue = gpu
    for lg in range(1, 11):
        n = 1 << lg
        x = k4[f"x_{n}"]
        y, which, _ = _run(fw, dev, queue, "Forward", x, n, path=path, small_reg=small_reg)
        _check(oracle, y, k4[f"fwd_{n}"], n)
        assert which == lg % 2
        y, _, _ = _run(fw, dev, queue, "Onlyinverse", x, n, path=path, small_reg=small_reg)
        _check(oracle, y, k4[f"inv_unscaled_{n}"], n)
        y, _, _ = _run(fw, dev, queue, "Inverse", x, n, path=path, small_reg=small_reg)
        _check(oracle, y, k4[f"inv_unscaled_{n}"] / n, n)


def test_literal_recurrence_matches_restatement_bitwise_shape(gpu, oracle):
    """path=2 is the reference recurrence one launch per stage (fft.wgsl:27-62); its distance to the
    CPU restatement of the same recurrence is pure fma-contraction noise."""
    fw, dev, queue = gpu
    n, batch = 4096, 3
    x = oracle.gen_input(n, batch)
    y, _, _ = _run(fw, dev, queue, "Forward", x, n, path=2)
    yr, _ = oracle.forward_ref(x, n)
    d = np.abs(y.astype(np.complex128) - yr.astype(np.complex128)).max() / np.abs(yr).max()
    assert d <= 2e-6, d


# ---- size sweep incl. ragged batches, 64-bit-free small cases ----
@pytest.mark.parametrize("lg,batch", [(0, 5), (1, 7), (2, 1), (3, 1000), (4, 1), (5, 33), (6, 129), (9, 2500),
                                      (10, 1), (11, 5), (12, 3), (13, 2), (14, 3), (15, 3), (16, 2), (17, 5), (18, 1),
                                      (19, 3), (21, 1), (22, 3), (23, 1), (25, 1)])
def test_size_sweep(gpu, oracle, lg, batch):
    fw, dev, queue = gpu
    n = 1 << lg
    x = oracle.gen_input(n, batch, first_transform=lg)
    y, which, _ = _run(fw, dev, queue, "Forward", x, n)
    assert which == lg % 2
    _check(oracle, y, oracle.dft_f64(x, n, -1), n)
    # K5: forward then scaled inverse is the identity
    z, _, _ = _run(fw, dev, queue, "Inverse", y, n)
    _check(oracle, z, x.astype(np.complex128), n)


# ---- C1 / C2 / C5 shapes ----
def test_config_c1_n1024_batch1(gpu, oracle):
    fw, dev, queue = gpu
    x = oracle.gen_input(1024, 1)
    y, which, _ = _run(fw, dev, queue, "Forward", x, 1024)
    assert which == 0
    mx, l2 = _check(oracle, y, oracle.dft_f64(x, 1024, -1), 1024)
    yr, _ = oracle.forward_ref(x, 1024)
    print("C1 max_rel %.3g rel_l2 %.3g; vs fp32 restatement %.3g" % (
        mx, l2, np.abs(y - yr).max() / np.abs(yr).max()))


# fused in-place pipeline (path 5, the default): (batch, depth, workgroups); two-launch ring (path 1): (batch, group, streams)
# (two-launch ring is the default since the fused path's in-launch waits cost more than they save: DESIGN.md)
@pytest.mark.parametrize("batch,path,a,b", [(1, 5, 4, 512), (3, 5, 1, 512), (5, 5, 2, 64), (17, 5, 4, 512),
                                            (40, 5, 6, 300), (9, 5, 16, 512),
                                            (1, 1, 8, 2), (3, 1, 2, 2), (5, 1, 2, 1), (17, 1, 4, 3),
                                            (1, 10, 8, 2), (3, 10, 2, 2), (5, 10, 2, 1), (17, 10, 4, 3), (23, 10, 3, 2)])
def test_config_c2_n1m(gpu, oracle, batch, path, a, b):
    fw, dev, queue = gpu
    n = 1 << 20
    x = oracle.gen_input(n, batch)
    # path 10 = two-launch ring without the mixed launches (mix=0); path 1 = mixed launches (default)
    kw = (dict(path=5, depth=a, wgs=b) if path == 5 else
          dict(path=1, group=a, streams=b, mix=1 if path == 1 else 0))
    y, which, plan = _run(fw, dev, queue, "Forward", x, n, **kw)
    assert which == 0 and plan.get("path") == (5 if path == 5 else 1)
    r = oracle.dft_f64(x, n, -1)
    mx, l2 = _check(oracle, y, r, n)
    print("C2 batch %d: max_rel %.3g rel_l2 %.3g" % (batch, mx, l2))
    # K8 batch independence: transform b of the batch == the same data run alone
    if batch > 1:
        b = batch - 1
        y1, _, _ = _run(fw, dev, queue, "Forward", x[b * n:(b + 1) * n], n)
        assert np.array_equal(y1.view(np.uint32), y[b * n:(b + 1) * n].view(np.uint32))
    # inverse family on the fast path
    z, _, _ = _run(fw, dev, queue, "Inverse", y, n, **kw)
    _check(oracle, z, x.astype(np.complex128), n)


@pytest.mark.parametrize("policy", [0, 1, 2, 3, 4, 5, 6, 7])
def test_n1m_cache_policies_are_bit_identical(gpu, oracle, policy):
    """Cache-policy variants (write-through / non-temporal accesses, with or without fences) change how
    workgroups hand the intermediate over, never the arithmetic: every variant of both pipelines must
    reproduce the default two-launch result bit for bit, also when the buffer is much larger than L2 and
    execs run back to back (stale-line hazards show up as mismatching 128-byte lines)."""
    fw, dev, queue = gpu
    n, batch = 1 << 20, 96                      # 768 MiB: far beyond L2 (32 MiB) and Infinity Cache (256 MiB)
    x = oracle.gen_input(n, batch, first_transform=5)
    ref, _, _ = _run(fw, dev, queue, "Forward", x, n, path=1, policy=0)
    mx, _ = oracle.compare(ref[:n], oracle.dft_f64(x[:n], n, -1))
    assert mx <= REL_TOL
    for kw in (dict(path=5, depth=4), dict(path=5, depth=2, wgs=200), dict(path=1, group=8, streams=2, mix=0),
               dict(path=1, group=8, streams=2, mix=1), dict(path=1, group=5, streams=3, mix=1)):
        for rep in range(2):
            y, _, _ = _run(fw, dev, queue, "Forward", x, n, policy=policy, **kw)
            bad = np.flatnonzero(y.view(np.uint64) != ref.view(np.uint64))
            assert bad.size == 0, (policy, kw, rep, bad.size, bad[:8])


def test_n1m_matches_literal_recurrence(gpu, oracle):
    fw, dev, queue = gpu
    n = 1 << 20
    x = oracle.gen_input(n, 2, first_transform=11)
    y_fast, _, _ = _run(fw, dev, queue, "Forward", x, n)
    y_two, _, _ = _run(fw, dev, queue, "Forward", x, n, path=5)
    y_lit, _, _ = _run(fw, dev, queue, "Forward", x, n, path=2)
    d = np.abs(y_fast.astype(np.complex128) - y_lit).max() / np.abs(y_lit).max()
    assert d <= REL_TOL, d
    # fused in-place (path 5) and two-launch (default) pipelines run the same arithmetic: bit-identical
    assert np.array_equal(y_fast.view(np.uint32), y_two.view(np.uint32))


def test_config_c5_n16m_batch1(gpu, oracle):
    fw, dev, queue = gpu
    n = 1 << 24
    x = oracle.gen_input(n, 1)
    y, which, plan = _run(fw, dev, queue, "Forward", x, n)
    assert which == 0 and plan.get("path") == 7  # tiled: 256 x 256 x 256, three k_tile16 passes
    mx, l2 = _check(oracle, y, oracle.dft_f64(x, n, -1), n)
    print("C5 max_rel %.3g rel_l2 %.3g" % (mx, l2))


# ---- properties ----
def test_impulse_tone_parseval_linearity(gpu, oracle):
    fw, dev, queue = gpu
    for n in (256, 1 << 20):
        k = np.arange(n)
        p, q = 5, 37
        x = np.zeros(n, np.complex64); x[p] = 1
        y, _, _ = _run(fw, dev, queue, "Forward", x, n)
        _check(oracle, y, np.exp(-2j * np.pi * ((p * k) % n) / n), n)            # K2
        x = np.exp(2j * np.pi * ((q * k) % n) / n).astype(np.complex64)
        y, _, _ = _run(fw, dev, queue, "Forward", x, n)
        e = np.zeros(n, np.complex128); e[q] = n
        mx, _ = oracle.compare(y, e)
        assert mx <= REL_TOL                                                      # K3
        a = oracle.gen_input(n, 1, first_transform=1)
        b = oracle.gen_input(n, 1, first_transform=2)
        ya, _, _ = _run(fw, dev, queue, "Forward", a, n)
        yb, _, _ = _run(fw, dev, queue, "Forward", b, n)
        yab, _, _ = _run(fw, dev, queue, "Forward", (a + 2 * b).astype(np.complex64), n)
        mx, _ = oracle.compare(yab, ya.astype(np.complex128) + 2 * yb.astype(np.complex128))
        assert mx <= REL_TOL                                                      # linearity
        ea = np.sum(np.abs(a.astype(np.complex128)) ** 2)
        eya = np.sum(np.abs(ya.astype(np.complex128)) ** 2) / n
        assert abs(ea - eya) <= 1e-5 * ea                                         # K7 Parseval


def test_onlyinverse_plus_normalize_equals_inverse(gpu, oracle):
    fw, dev, queue = gpu
    for n, batch in ((512, 40), (1 << 20, 2)):
        x = oracle.gen_input(n, batch)
        src = _upload(fw, dev, queue, x)
        src2 = dev.create_buffer(x.nbytes)
        oi = fw.Onlyinverse(dev, queue, src, src2, n)
        nm = fw.Normalize(dev, queue, src, src2, n)
        enc = dev.create_command_encoder()
        oi.proc(enc)
        out = nm.proc(enc)
        y2 = out.map_read(stream=enc)
        y1, _, _ = _run(fw, dev, queue, "Inverse", x, n)
        assert np.array_equal(y1.view(np.uint32), y2.view(np.uint32))             # K6, bit for bit


def test_device_generator_is_bit_identical_to_oracle(gpu, oracle):
    fw, dev, queue = gpu
    n, batch = 4096, 9
    buf = dev.create_buffer(n * batch * 8)
    dev.fill_synthetic(buf, n, first_transform=3, scale=2.0 ** -7)
    dev.poll()
    got = buf.map_read()
    want = oracle.gen_input(n, batch, first_transform=3, scale=2.0 ** -7)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---- C3: the headline shape, full size, 64-bit offsets ----
def test_config_c3_full_size_sampled(gpu, oracle):
    fw, dev, queue = gpu
    n, batch = 1 << 20, 4096
    info = dev.info()
    if info["hbm_bytes"] < 48 * 2 ** 30:
        pytest.skip("needs a 32 GiB buffer")
    buf = dev.create_buffer(n * batch * 8)            # 32 GiB, element offsets exceed 2^32
    dev.fill_synthetic(buf, n)
    plan = fw.Forward(dev, queue, buf, n)
    enc = dev.create_command_encoder()
    out = plan.proc(enc)
    enc.synchronize()
    assert out is buf and plan.get("device_error") == 0
    rng = np.random.default_rng(7)
    sample = [0, batch - 1] + sorted(rng.choice(np.arange(1, batch - 1), 14, replace=False).tolist())
    worst = 0.0
    for t in sample:
        y = out.map_read(offset=t * n * 8, size=n * 8, stream=enc)
        x = oracle.gen_input(n, 1, first_transform=t)
        mx, l2 = oracle.compare(y, oracle.dft_f64(x, n, -1))
        assert mx <= REL_TOL and l2 <= REL_TOL, (t, mx, l2)
        worst = max(worst, mx)
    print("C3 sampled transforms %s worst max_rel %.3g" % (sample, worst))
    # size-independent property at full size: forward then scaled inverse restores the generator output
    inv = fw.Inverse(dev, queue, buf, n)
    out2 = inv.proc(enc)
    enc.synchronize()
    for t in (0, 2049, batch - 1):
        z = out2.map_read(offset=t * n * 8, size=n * 8, stream=enc)
        x = oracle.gen_input(n, 1, first_transform=t)
        mx, _ = oracle.compare(z, x.astype(np.complex128))
        assert mx <= REL_TOL


# ---- error behaviour of the boundary ----
def test_rejects_bad_arguments(gpu):
    fw, dev, queue = gpu
    buf = dev.create_buffer(8 * 1000)
    with pytest.raises(fw.FwaError) as e:
        fw.Forward(dev, queue, buf, 1000)            # not a power of two
    assert e.value.status == 1
    with pytest.raises(fw.FwaError):
        fw.Forward(dev, queue, buf, 16)              # 1000 % 16 != 0
    with pytest.raises(fw.FwaError):
        fw.Forward(dev, queue, buf, 0)
    ok = dev.create_buffer(8 * 1024)
    small = dev.create_buffer(8 * 512)
    with pytest.raises(fw.FwaError):
        fw.Onlyinverse(dev, queue, ok, small, 512)   # size mismatch
    with pytest.raises(fw.FwaError):
        fw.Onlyinverse(dev, queue, ok, ok, 512)      # same buffer twice
    empty = dev.create_buffer(0)
    p = fw.Forward(dev, queue, empty, 1024)           # empty batch is legal and a no-op
    assert p.get("batch") == 0
    enc = dev.create_command_encoder()
    assert p.proc(enc) is empty


def test_cpp_mirror_replays_reference_example(gpu, tmp_path):
    """include/fft_wgpu.hpp: C++ replay of examples/basic_inverse2.rs (Onlyinverse + Normalize, n=512,
    constant input, max error < 1e-5) through the C ABI."""
    import os
    import subprocess
    from conftest import ROOT
    exe = tmp_path / "example"
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "example_basic_inverse2.cpp"),
                           "-L" + os.path.join(ROOT, "fft_wgpu_amd"), "-lfft_wgpu_amd", "-o", str(exe)])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "fft_wgpu_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "max error 0" in r.stdout


def test_split_path_cross_check(gpu, oracle):
    """FWA_FORCE_SPLIT=1 selects the older split decomposition (strided radix passes + sub-transforms +
    permute) for 2^15..2^30; both decompositions must agree with the fp64 DFT."""
    import os
    fw, dev, queue = gpu
    os.environ["FWA_FORCE_SPLIT"] = "1"
    try:
        for lg, batch in ((15, 3), (18, 2), (21, 1), (24, 1)):
            n = 1 << lg
            x = oracle.gen_input(n, batch, first_transform=lg)
            y, which, plan = _run(fw, dev, queue, "Forward", x, n)
            assert plan.get("path") == 6 and which == lg % 2
            _check(oracle, y, oracle.dft_f64(x, n, -1), n)
    finally:
        del os.environ["FWA_FORCE_SPLIT"]


def test_plan_owned_result_buffer_outlives_temporary_plan(gpu, oracle):
    """Forward/Inverse with odd log2 n return a view of the plan's own second buffer (processor.rs:13,153-157);
    the view must keep the plan alive, as the Rust borrow does."""
    import gc
    fw, dev, queue = gpu
    x = oracle.gen_input(512, 4)
    src = _upload(fw, dev, queue, x)
    enc = dev.create_command_encoder()
    out = fw.Forward(dev, queue, src, 512).proc(enc)   # the plan object is a temporary
    gc.collect()
    y = out.map_read(stream=enc)
    _check(oracle, y, oracle.dft_f64(x, 512, -1), 512)


def test_large_single_transform_properties(gpu):
    """n = 2^27 (1 GiB per transform, three 512-point passes, 64-bit offsets inside ONE transform): no CPU FFT of
    that size; size-independent properties instead -- an impulse at p transforms to exp(-2*pi*i*p*k/n) (every
    output checked), and forward followed by the scaled inverse restores the input bit pattern to <= 1e-5."""
    fw, dev, queue = gpu
    lg = 27
    n = 1 << lg
    if dev.info()["hbm_bytes"] < 16 * 2 ** 30:
        pytest.skip("needs ~6 GiB of device memory")
    p = 3 * 5 * 7 * 11 * 13 + 2 ** 20
    x = np.zeros(n, dtype=np.complex64)
    x[p] = 1
    src = _upload(fw, dev, queue, x)
    plan = fw.Forward(dev, queue, src, n)
    assert plan.get("path") == 7
    enc = dev.create_command_encoder()
    out = plan.proc(enc)
    y = out.map_read(stream=enc)
    assert (out.device_ptr == src.device_ptr) == (lg % 2 == 0)
    k = np.arange(n, dtype=np.int64)
    ph = ((p * k) % n).astype(np.float64) * (-2.0 * np.pi / n)
    err = max(np.abs(y.real - np.cos(ph)).max(), np.abs(y.imag - np.sin(ph)).max())
    assert err <= REL_TOL, err
    inv = fw.Inverse(dev, queue, out, n)
    z = inv.proc(enc).map_read(stream=enc)
    assert np.abs(z - x).max() <= REL_TOL


def test_seeded_fuzz_over_sizes_kinds_and_batches(gpu, oracle):
    """60 seeded random (n, batch, plan kind) combinations, ragged batches included, against the fp64 DFT."""
    fw, dev, queue = gpu
    rng = np.random.default_rng(20251004)
    for case in range(60):
        lg = int(rng.integers(0, 19))
        n = 1 << lg
        batch = int(rng.integers(1, max(2, min(3000, (1 << 21) >> lg) + 1)))
        kind = ("Forward", "Inverse", "Onlyinverse")[int(rng.integers(0, 3))]
        x = oracle.gen_input(n, batch, first_transform=case)
        y, which, _ = _run(fw, dev, queue, kind, x, n)
        assert which == lg % 2, (case, lg, batch, kind)
        r = oracle.dft_f64(x, n, -1 if kind == "Forward" else +1)
        if kind == "Inverse":
            r = r / n
        for t in (0, batch // 2, batch - 1):
            mx, l2 = oracle.compare(y[t * n:(t + 1) * n], r[t * n:(t + 1) * n])
            assert mx <= REL_TOL and l2 <= REL_TOL, (case, lg, batch, kind, t, mx, l2)
