"""agile_grasp2_amd -- MI355X-native hot path of agile_grasp2 behind a C-ABI (libag2hip.so).

The product is the HIP library in agile_grasp2_amd/csrc (C-ABI: include/ag2_c.h) and the C++ host
mirror of the reference API in include/agile_grasp2/.  The Python here is the harness side only:
ctypes bindings for tests/bench (capi), synthetic scenes (scene) and seeded weights (weights).
"""
