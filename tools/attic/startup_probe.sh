#!/bin/bash
# where the CLI's start-up time goes: process start + library load (ld.so), device bring-up, first allocations
cd "$(dirname "$0")/.."
for i in 1 2 3; do /usr/bin/time -f "info wall %e s" oswald_amd/oswald -O info > /dev/null; done
LD_DEBUG=statistics oswald_amd/oswald -O info 2>&1 | grep -E "total startup|relocation|load" | head
python - <<'P'
import ctypes, time
t=time.time(); ctypes.CDLL("/opt/rocm/lib/libamdhip64.so"); print("dlopen libamdhip64 %.1f ms" % ((time.time()-t)*1e3))
t=time.time(); ctypes.CDLL("/opt/rocm/lib/librccl.so.1"); print("dlopen librccl %.1f ms" % ((time.time()-t)*1e3))
t=time.time(); l=ctypes.CDLL("oswald_amd/liboswald_hip.so"); print("dlopen liboswald_hip %.1f ms" % ((time.time()-t)*1e3))
n=ctypes.c_int(0); t=time.time(); l.oswald_hip_device_count(ctypes.byref(n)); print("device_count %.1f ms" % ((time.time()-t)*1e3))
h=ctypes.c_void_p(); t=time.time(); l.oswald_hip_init(1, None, ctypes.byref(h)); print("init %.1f ms" % ((time.time()-t)*1e3))
o=ctypes.c_uint64(0); t=time.time(); l.oswald_hip_max_chunk_size(h, 0, 20, 65520, ctypes.byref(o)); print("max_chunk_size %.1f ms -> %d" % ((time.time()-t)*1e3, o.value))
t=time.time(); l.oswald_hip_reserve(h, -1, 3200); print("reserve %.1f ms" % ((time.time()-t)*1e3))
P
