"""dev tool: register / scratch / LDS figures of every kernel in a built librcs_hip.so (reads the gfx950 code object's notes).
usage: so_regs.py lib.so [kernel-name-substring]"""
import re, subprocess, sys, tempfile, os

so = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
LL = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([f"{LL}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    co = os.path.join(d, "gfx950.co")
    subprocess.check_call([f"{LL}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"])
    notes = subprocess.check_output([f"{LL}/llvm-readelf", "--notes", co], text=True)
    if len(sys.argv) > 3:
        subprocess.check_call(["cp", co, sys.argv[3]])
for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size", notes, re.S):
    body = m.group(0)
    f = dict(re.findall(r"\.(\w+):\s+(\S+)", body))
    name = subprocess.check_output(["c++filt", f.get("name", "?")], text=True).strip()
    if pat in name:
        print(name[:120], "| vgpr", f.get("vgpr_count"), "agpr", f.get("agpr_count"), "sgpr", f.get("sgpr_count"), "scratch", f.get("private_segment_fixed_size"),
              "lds", f.get("group_segment_fixed_size"), "spill v/s", f.get("vgpr_spill_count"), f.get("sgpr_spill_count"))
