"""oswald_amd -- MI355X-native Smith-Waterman protein database search path.

A from-scratch replacement for the accelerator side of enzorucci/OSWALD
(`device/sw.cl` + the OpenCL enqueue path of `host/src/FPGAsearch.c`): hand
written HIP kernels for gfx950 behind the C ABI of include/oswald_hip.h.
The Python package is plumbing around that library (ctypes binding, synthetic
workloads, host-side layout mirror); the compute path is liboswald_hip.so and
nothing else -- there is no CPU fallback.
"""
__version__ = "0.1.0"
