"""per-kernel register / scratch / code-size summary of a device assembly listing (dev aid)
usage: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Icsrc -S --cuda-device-only x.hip -o x.s ; isa_stats.py x.s [substr]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
names = re.findall(r"^\s*\.amdhsa_kernel (\S+)", txt, re.M)
for nm in names:
    def get(key):
        m = re.search(r"\.set " + re.escape(nm) + r"\." + key + r", (\d+)", txt)
        return int(m.group(1)) if m else -1
    m = re.search(r"^" + re.escape(nm) + r":.*?\n(.*?)\.Lfunc_end\d+:", txt, re.S | re.M)
    body = m.group(1) if m else ""
    ins = [l for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    if sub and sub not in dem.replace(" ", ""):
        continue
    cnt = lambda p: sum(1 for l in ins if re.match(r"\s*" + p, l))
    print(f"{dem}: vgpr {get('num_vgpr')} sgpr {get('numbered_sgpr')} scratch {get('private_seg_size')} instrs {len(ins)} "
          f"ds_read {cnt('ds_read')} ds_write {cnt('ds_write')} gload {cnt('global_load')} gstore {cnt('global_store')} barrier {cnt('s_barrier')} waitcnt {cnt('s_waitcnt')}")
