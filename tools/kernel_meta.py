"""Code-object metadata of the shipped library: VGPR / AGPR / spill counts, scratch and LDS bytes of every kernel
(llvm-objdump --offloading + llvm-readelf --notes). usage: python tools/kernel_meta.py [substring ...] [--csv]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "city-rollup_amd", "libcityprover_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(so=SO):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(so, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
        rows = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
                e = ".agpr_count" + e
                g = lambda k: int((re.search(r"\." + k + r":\s+(\d+)", e) or [0, 0])[1])
                m = re.search(r"\.name:\s+(\S+)", e)
                if m:
                    rows.append(dict(symbol=m.group(1), vgpr=g("vgpr_count"), agpr=g("agpr_count"), vgpr_spill=g("vgpr_spill_count"),
                                     sgpr=g("sgpr_count"), sgpr_spill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"),
                                     lds=g("group_segment_fixed_size"), max_wg=g("max_flat_workgroup_size")))
        names = subprocess.run(["c++filt"], input="\n".join(r["symbol"] for r in rows), capture_output=True, text=True).stdout.split("\n")
        for r, n in zip(rows, names):
            r["name"] = re.sub(r"\(.*$", "", n)
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    pats = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = [r for r in kernels() if not pats or any(p in r["name"] for p in pats)]
    if "--csv" in sys.argv:
        print("kernel,vgpr,agpr,vgpr_spill,sgpr,scratch_bytes,lds_bytes")
        for r in rows:
            print(f'"{r["name"]}",{r["vgpr"]},{r["agpr"]},{r["vgpr_spill"]},{r["sgpr"]},{r["scratch"]},{r["lds"]}')
    else:
        for r in rows:
            print(f'{r["name"][:90]:90s} vgpr {r["vgpr"]:3d} agpr {r["agpr"]:3d} spill {r["vgpr_spill"]:4d} scratch {r["scratch"]:5d} lds {r["lds"]:6d}')
