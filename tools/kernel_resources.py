#!/usr/bin/env python3
"""Register / LDS / scratch use of the kernels of one HIP source, as the compiler reports it.

    python tools/kernel_resources.py apply_fwd_seg.hip [--tools] [pattern ...]

Compiles the file for the device only with -Rpass-analysis=kernel-resource-usage (same flags as
hdrnet_amd/build.py) and prints one line per kernel whose demangled name contains every pattern.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hdrnet_amd import build as hb  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    src, pats = args[0], args[1:]
    extra = dict(hb.SOURCES + hb.TOOLS_ONLY_SOURCES)[src]
    define = ["-DHDRNET_TOOLS_BUILD"] if "--tools" in sys.argv else []
    cmd = [hb.hipcc(), *hb.COMMON, *define, *extra, "-I", hb.CSRC, "-I", hb.INCLUDE, "--offload-device-only",
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(hb.CSRC, src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True,
                           text=True).stdout.splitlines()
    for (name, f), dn in zip(rows.items(), names):
        dn = re.sub(r"^void hdrnet_amd::\(anonymous namespace\)::", "", dn)
        dn = re.sub(r"\(.*$", "", dn)
        if all(p in dn for p in pats):
            print(f"vgpr {f.get('VGPRs', '?'):>3s} agpr {f.get('AGPRs', '?'):>3s} sgpr {f.get('SGPRs', '?'):>3s} "
                  f"occ {f.get('Occupancy', '?'):>2s} lds {f.get('LDS Size', '?'):>6s} scratch {f.get('ScratchSize', '?'):>4s}  {dn}")


if __name__ == "__main__":
    main()
