"""CPU oracles for the DCLL hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
package (snn_modulation_classification_amd) never does.
  torch_ref.py    same eager torch op sequence as the reference (bit-identical to it under the same torch build)
  dcll_oracle.c   plain C with the pinned fmaf-chain order of include/dcll_hip.h (what the HIP kernels must match
                  bit for bit); c_oracle.py is its ctypes binding
"""
