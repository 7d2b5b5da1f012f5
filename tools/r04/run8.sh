#!/bin/bash
python - <<'P'
import ctypes, glob
hip = ctypes.CDLL("libamdhip64.so")
b = ctypes.create_string_buffer(64)
print("rc", hip.hipDeviceGetPCIBusId(b, 64, 0), b.value)
bdf = b.value.decode().lower()
for p in ["/sys/bus/pci/devices/%s/numa_node" % bdf, "/sys/bus/pci/devices/%s/local_cpulist" % bdf]:
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, e)
print(glob.glob("/sys/devices/system/node/node*"))
P
echo "--- taskset socket 0"; taskset -c 0-63 tools/experiments/d2h_bw2 | head -4
echo "--- taskset socket 1"; taskset -c 64-127 tools/experiments/d2h_bw2 | head -4
