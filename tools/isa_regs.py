"""dev tool: register / scratch / LDS figures of the kernels of an ISA listing (hipcc -S --cuda-device-only).
usage: isa_regs.py file.s [kernel-name-substring]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", s, re.S):
    if pat not in m.group(1):
        continue
    f = dict(re.findall(r"\.amdhsa_(\w+) (\S+)", m.group(2)))
    print(m.group(1)[:110], "vgpr+agpr", f.get("next_free_vgpr"), "accum_offset", f.get("accum_offset"), "sgpr", f.get("next_free_sgpr"),
          "scratch", f.get("private_segment_fixed_size"), "lds", f.get("group_segment_fixed_size"))
