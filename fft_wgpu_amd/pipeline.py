"""The reference's benchmark loop (src/examples/basic.rs:72-127: write_buffer -> proc -> copy_buffer_to_buffer ->
map/read back, every iteration) as a pipeline: pinned staging, multi-buffered device buffers, two HIP streams --
A: upload + transform + device copy, B: read-back -- so the read-back of iteration i overlaps the upload of i+1
(the host link is full duplex).  Host-link bound by design: this is the PCIe-inclusive figure DESIGN.md quotes next
to (never instead of) the device-resident throughput.

Structure, chosen by measurement (tools/archive/pipe_probe.py, profiles/round2/host_pipeline_probe.jsonl; N = 512 x 2500 =
10.24 MB each way per iteration; the link alone carries 53 GB/s one way, 47 GB/s each way with both directions busy):
  * one device-side dependency per iteration (A -> B, an event recorded after the device copy): 0.23 ms / iteration;
  * guarding slot reuse with a second device-side dependency (B -> A) costs 2x on this stack (0.46 ms): the runtime
    serialises SDMA copies behind cross-queue waits; the guard is a HOST wait on the slot's read-back event instead,
    which the caller needs anyway before it touches the result;
  * a third stream for the upload, or copy kernels over the mapped pinned arrays instead of SDMA, are no faster.
"""
import numpy as np

from .device import Event


class HostPipeline:
    """plan_factory(device, queue, src_buffer) -> plan; `submit` feeds one batch, `result` returns a finished one."""

    def __init__(self, device, queue, plan_factory, n_samples, slots=2):
        self.device, self.queue = device, queue
        self.nbytes = n_samples * 8
        self.slots = slots
        self.ex = device.create_command_encoder()    # stream A: upload, transform, copy to staging
        self.down = device.create_command_encoder()  # stream B: read-back
        self.hin = [device.pinned_array(n_samples) for _ in range(slots)]
        self.hout = [device.pinned_array(n_samples) for _ in range(slots)]
        self.src = [device.create_buffer(self.nbytes) for _ in range(slots)]
        self.staging = [device.create_buffer(self.nbytes) for _ in range(slots)]
        self.plans = [plan_factory(device, queue, b) for b in self.src]
        self.e_ex = [Event(device) for _ in range(slots)]
        self.e_down = [Event(device) for _ in range(slots)]
        self.it = 0

    def submit(self, data=None):
        """Enqueue one iteration.  Blocks only until the slot's previous read-back (`slots` iterations ago) has
        finished; `data` (complex64, n_samples) is then copied into the slot's pinned input (None re-sends what the
        slot holds).  Returns the slot index."""
        s = self.it % self.slots
        if self.it >= self.slots:
            self.e_down[s].synchronize()   # hout[s] / staging[s] / src[s] / hin[s] of iteration it - slots are free
        if data is not None:
            np.copyto(self.hin[s], data)
        self.queue.write_buffer(self.src[s], 0, self.hin[s], encoder=self.ex)            # basic.rs:73
        out = self.plans[s].proc(self.ex)                                                 # basic.rs:79
        self.ex.copy_buffer_to_buffer(out, 0, self.staging[s], 0, self.nbytes)            # basic.rs:84-90
        self.e_ex[s].record(self.ex)
        self.down.wait_event(self.e_ex[s])
        self.device.download_async(self.hout[s], self.staging[s], self.down)             # basic.rs:105-122
        self.e_down[s].record(self.down)
        self.it += 1
        return s

    def result(self, slot):
        """Wait for the read-back of `slot` and return the pinned result array (valid until the slot is reused)."""
        self.e_down[slot].synchronize()
        return self.hout[slot]

    def drain(self):
        self.ex.synchronize()
        self.down.synchronize()
