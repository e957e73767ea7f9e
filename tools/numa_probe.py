"""Where does this process run relative to the GPU?  Prints the affinity mask, the NUMA layout, the GPU's local CPU list and,
for a short run of the headline step, the CPU the host thread was on and the host loop's time per step."""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psutil, torch
print("affinity:", len(os.sched_getaffinity(0)), "cpus", sorted(os.sched_getaffinity(0))[:4], "...", sorted(os.sched_getaffinity(0))[-4:])
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(n.split("/")[-2], open(n).read().strip())
bus = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
print("gpu pci bus id:", bus, getattr(torch.cuda.get_device_properties(0), "pci_device_id", None))
for d in glob.glob("/sys/bus/pci/devices/*/numa_node"):
    try:
        cls = open(os.path.dirname(d) + "/class").read().strip()
        ven = open(os.path.dirname(d) + "/vendor").read().strip()
    except OSError:
        continue
    if ven == "0x1002" and cls.startswith(("0x0302", "0x0380", "0x0300", "0x1200")):
        print(os.path.dirname(d).split("/")[-1], "class", cls, "numa_node", open(d).read().strip(), "local_cpulist", open(os.path.dirname(d) + "/local_cpulist").read().strip())
print("loadavg:", open("/proc/loadavg").read().strip())
import bench
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
pc = make_scene(1_000_000, seed=0); pc.train()
params = [p for p in pc.parameters() if p.requires_grad]
me = psutil.Process()
for i in range(40):
    torch.cuda.synchronize(); t = time.perf_counter()
    bench.one_step(pc, cams[i % 8], pipe, bg, w, 20000, params, None)
    th = time.perf_counter() - t
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    if i % 4 == 3:
        print(f"step {i:2d}: {dt * 1e3:6.2f} ms (host loop {th * 1e3:5.2f}) on cpu {me.cpu_num()}", flush=True)
print("loadavg:", open("/proc/loadavg").read().strip())
