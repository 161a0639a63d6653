#!/usr/bin/env python3
"""Host-to-device copy rate of a pageable 0.5 GB array as the process is placed (default affinity) and with the process bound to the CPUs of the
GPU's own NUMA node (memory allocated after binding).  Some boxes of the pool upload at 13 GB/s instead of 52: is it the placement?"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

def bw(tag):
    x = np.ones((100000, 5120), dtype=np.uint8)          # fresh pages, touched by this thread
    d = torch.empty((100000, 5120), dtype=torch.uint8, device="cuda")
    t_host = torch.from_numpy(x)
    d.copy_(t_host); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); d.copy_(t_host); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"{tag}: {x.nbytes / min(ts) / 1e9:.1f} GB/s  ({min(ts) * 1e3:.1f} ms)", flush=True)

props = torch.cuda.get_device_properties(0)
bdf = None
try:
    bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
except Exception as e:
    print("no pci ids from torch:", e)
print("device", props.name, "pci", bdf)
node, cpus = None, None
if bdf and os.path.exists(f"/sys/bus/pci/devices/{bdf}/numa_node"):
    node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
    cpus = open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip()
print("gpu numa node", node, "local cpus", cpus)
print("process affinity:", len(os.sched_getaffinity(0)), "cpus")
for f in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(f.split("/")[-2], open(f).read().strip())
bw("default placement")
def parse(cl):
    out = set()
    for part in cl.split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out
allowed = os.sched_getaffinity(0)
for f in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    name = f.split("/")[-2]
    want = parse(open(f).read().strip()) & allowed
    if want:
        os.sched_setaffinity(0, want)
        bw(f"bound to {name}" + (" (the GPU's node)" if node is not None and name == f"node{node}" else " (remote)"))
os.sched_setaffinity(0, allowed)
