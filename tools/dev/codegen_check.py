"""development: per-step instruction mix of a packed kernel's main loop (catches register-copy blow-ups)
usage: codegen_check.py <method nw|ga|sw> <K> [G = 8|16]"""
import re, subprocess, sys, collections
m, k = sys.argv[1], int(sys.argv[2])
mi = {"nw": 0, "ga": 1, "sw": 2}[m]
src = f"/root/repo/sequencealigner_amd/csrc/sa_systolic_pk_{m}.hip"
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-x", "hip", "--cuda-device-only",
                      "-S", src, "-o", "-"], capture_output=True, text=True).stdout
g = int(sys.argv[3]) if len(sys.argv) > 3 else 8
f16 = (sys.argv[4] == "f16") if len(sys.argv) > 4 else g == 8
name = f"_ZN12_GLOBAL__N_116sa_k_systolic_pkILi{mi}ELi{g}ELi{k}ELb{int(f16)}EEEv9SaSysArgs"
body = asm[asm.index(name + ":"):]
body = body[:body.index("s_endpgm")]
L = body.split("\n")
idx = [i for i, l in enumerate(L) if re.search(r"s_bitcmp[01]_b32 s\d+, \d+$", l)]
for a, b in list(zip(idx, idx[1:]))[:16]:
    seg = [l.split()[0] for l in L[a:b] if l.strip() and not l.strip().startswith(";") and not l.strip().startswith(".")]
    c = collections.Counter(seg)
    print(L[a].strip(), "instrs", len(seg), "v_mov", c.get("v_mov_b32_e32", 0), "pk_max", c.get("v_pk_max_u16", 0), "add/sub", c.get("v_add_u32_e32", 0) + c.get("v_sub_u32_e32", 0),
          "dpp", c.get("v_mov_b32_dpp", 0), "s_nop", c.get("s_nop", 0))
md = re.search(name + r".*?\.vgpr_count:\s+(\d+)", asm[asm.index(".amdhsa_kernel " + name):] if False else asm, re.S)
for mm in re.finditer(r"\.name:\s+" + name + r"\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", asm):
    print("vgpr_count", mm.group(1))
