"""Generates tests/golden/*.npz / *.json.  Run from the repo root:

    python tests/golden/make_golden.py

The reference (Rust + WGSL over wgpu) cannot run in this pipeline (no rustc,
no Vulkan loader), so no vector here comes from a reference
run.  What is stored:

* ``reference_known_answers.json`` -- the inputs and expected outputs of the
  reference's OWN asserting tests and benchmark inputs (constant vectors; the
  expected output of a constant c is n*c at bin 0 forward / c at bin 0 for the
  scaled inverse and exactly 0 elsewhere), with the file:line of each.
* ``k4_random.npz`` -- seeded uniform(-1,1) complex64 inputs for every power of
  two 2..1024 and their DFT computed by numpy.fft in float64 (independent of
  both the oracle and the HIP path).
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20250614)
    arrays = {}
    for lg in range(1, 11):
        n = 1 << lg
        x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
        arrays[f"x_{n}"] = x
        arrays[f"fwd_{n}"] = np.fft.fft(x.astype(np.complex128))
        arrays[f"inv_unscaled_{n}"] = np.fft.ifft(x.astype(np.complex128)) * n
    np.savez_compressed(os.path.join(HERE, "k4_random.npz"), **arrays)

    known = {
        "comment": "constant-input cases held by the reference's own tests/examples",
        "tolerance_abs": 1e-5,  # examples/basic_inverse.rs:250, basic_inverse2.rs:281
        "cases": [
            {"src": "src/examples/basic_inverse.rs:160,177,219-253", "plan": "Inverse",
             "n": 512, "c": [2.0, 42.0], "asserted_by_reference": True},
            {"src": "src/examples/basic_inverse2.rs:169,206-208,249-284", "plan": "Onlyinverse+Normalize",
             "n": 512, "c": [2.1327392395, 3.033729], "asserted_by_reference": True},
            {"src": "src/examples/basic.rs:167,201,250 (printed, not asserted)", "plan": "Forward",
             "n": 16, "c": [1.0, 0.0], "asserted_by_reference": False},
            {"src": "src/examples/basic.rs:32,66 (benchmark input)", "plan": "Forward",
             "n": 512, "c": [1.0, 0.0], "asserted_by_reference": False},
            {"src": "src/examples/basic_inverse.rs:33,67 (benchmark input)", "plan": "Inverse",
             "n": 512, "c": [2.0, 423.0], "asserted_by_reference": False},
            {"src": "src/examples/basic_inverse2.rs:33,76-78 (benchmark input)", "plan": "Onlyinverse+Normalize",
             "n": 512, "c": [1.0, 0.0], "asserted_by_reference": False},
        ],
    }
    with open(os.path.join(HERE, "reference_known_answers.json"), "w") as f:
        json.dump(known, f, indent=1)


if __name__ == "__main__":
    main()
