"""MI355X-native DCLL LIF timestep loop (drop-in for the hot path of ohjay/snn-modulation-classification).

  _lib      ctypes binding of libdcll_hip.so (include/dcll_hip.h) — no CPU fallback
  ops       tensor-level wrappers over the C ABI
  dcll      Conv2dDCLLlayer / DenseDCLLlayer / DCLLClassification with the reference's names and state-dict keys
  networks  YAML network builder (load_network_spec, ConvNetwork)
  data      spike encoders (iq2spiketrain, image2spiketrain)
  parallel  batch sharding over the GPUs of one node (RCCL tally all-reduce)
"""
__version__ = "0.1.0"
