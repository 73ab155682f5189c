"""development: instruction mix of a packed bundle kernel (catches register-copy blow-ups, SGPR spills and -- the one that
cost 9 % in round 3 -- FLAT loads in the main loop: every load must be global_load or s_load)
usage: codegen_check.py <method nw|ga|sw> [G = 8|16] [KLO = 9] [f16 = 1]"""
import collections, re, subprocess, sys
m = sys.argv[1] if len(sys.argv) > 1 else "nw"
g = int(sys.argv[2]) if len(sys.argv) > 2 else 8
klo = int(sys.argv[3]) if len(sys.argv) > 3 else (9 if g == 8 else 13)
f16 = int(sys.argv[4]) if len(sys.argv) > 4 else 1
mi = {"nw": 0, "ga": 1, "sw": 2}[m]
tu = f"sa_systolic_pk_{m}.hip" if g == 8 else f"sa_systolic_pk16{'hi' if klo >= 45 else ''}_{m}.hip"
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-x", "hip", "--cuda-device-only",
                      "-S", f"/root/repo/sequencealigner_amd/csrc/{tu}", "-o", "-"], capture_output=True, text=True).stdout
name = f"_ZN12_GLOBAL__N_123sa_k_systolic_pk_bundleILi{mi}ELi{g}ELi{klo}ELb{f16}EEEv9SaSysArgs"
body = asm[asm.index(name + ":"):]
body = body[:body.index("s_endpgm")]
ops = [l.split()[0] for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
c = collections.Counter(ops)
print(name)
print("instructions", len(ops))
for k in ("flat_load_ubyte", "flat_load_dword", "global_load_ubyte", "global_load_dword", "s_load_dword", "v_readlane_b32", "v_writelane_b32",
          "v_readfirstlane_b32", "v_pk_maximum3_f16", "v_pk_max_u16", "v_add_u32_e32", "v_sub_u32_e32", "v_mov_b32_e32", "v_mov_b32_dpp",
          "ds_read_b128", "ds_read_u16", "s_nop", "s_waitcnt", "s_setprio"):
    print(f"  {k:22s} {c.get(k, 0)}")
flat = sum(v for k, v in c.items() if k.startswith("flat_"))
print("FLAT memory instructions:", flat, "(must be 0)" if flat else "")
for key in ("vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count"):
    for mm in re.finditer(r"\.name:\s+" + name + r"\n(?:.*\n)*?\s+\." + key + r":\s+(\d+)", asm):
        print(key, mm.group(1))
