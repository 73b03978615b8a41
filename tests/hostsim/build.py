import os, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libhostsim.so")
def build():
    src = os.path.join(HERE, "hostsim.hip")
    csrc = os.path.join(HERE, "..", "..", "city-rollup_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if os.path.exists(SO) and all(os.path.getmtime(d) <= os.path.getmtime(SO) for d in deps):
        return SO
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC",
                    "-Wno-unused-value", "-o", SO, src], check=True)
    return SO
