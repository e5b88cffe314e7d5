"""CPU affinity of a measurement process to the NUMA node its GPU hangs off (see tools/broker_leg.py, profiles/r04_broker_numa.txt)."""
import os


def bind_to_gpu_numa_node(hip_device=0):
    """Run on the CPUs of the NUMA node the device hangs off, so that the buffers this process first-touches and pins are local to the
    device's PCIe root: DMA from the far socket of a two-socket host is slower, and which socket a fresh process lands on is luck - the
    same build read 11.9 and 18.1-18.3 GiB/s at 20 callers, unbound / bound, alternating on one box (profiles/r04_broker_numa.txt).  It is
    what a deployment does with `numactl --cpunodebind --membind` or the JVM's affinity.  The device's PCI address comes from the HIP runtime
    (hipDeviceGetPCIBusId), its node from sysfs; returns what was done (for the row)."""
    try:
        import ctypes
        import glob
        node = None
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(hip_device)) == 0:
                bdf = buf.value.decode().strip().lower()
                for cand in (bdf, "0000:" + bdf if bdf.count(":") == 1 else bdf):
                    path = "/sys/bus/pci/devices/%s/numa_node" % cand
                    if os.path.exists(path):
                        node = int(open(path).read().strip())
                        break
        except (OSError, AttributeError, ValueError):
            node = None
        if node is None:                                                 # no runtime at hand: the first AMD display device sysfs lists
            for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
                try:
                    if open(os.path.join(dev, "vendor")).read().strip() == "0x1002":
                        node = int(open(os.path.join(dev, "numa_node")).read().strip())
                        break
                except (OSError, ValueError):
                    continue
        if node is None or node < 0:
            return "no NUMA node reported for the GPU"
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        use = cpus & os.sched_getaffinity(0)
        if not use:
            return "GPU on NUMA node %d, none of its CPUs allowed here" % node
        os.sched_setaffinity(0, use)
        return "bound to %d CPUs of NUMA node %d (the GPU's)" % (len(use), node)
    except (OSError, ValueError, AttributeError) as e:
        return "not bound: %r" % (e,)


def bind_to_numa_node_of_pci(domain, bus, device):
    """The same, for a process that must not touch the HIP runtime by itself (bench.py: torch's bundled runtime is loaded, asking the
    system's libamdhip64 for the bus id would start a second runtime in the process): the PCI address comes from the caller."""
    try:
        path = "/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (int(domain), int(bus), int(device))
        node = int(open(path).read().strip())
        if node < 0:
            return "no NUMA node reported for the GPU"
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        use = cpus & os.sched_getaffinity(0)
        if not use:
            return "GPU on NUMA node %d, none of its CPUs allowed here" % node
        os.sched_setaffinity(0, use)
        return "bound to %d CPUs of NUMA node %d (the GPU's)" % (len(use), node)
    except (OSError, ValueError, AttributeError, TypeError) as e:
        return "not bound: %r" % (e,)
