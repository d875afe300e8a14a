"""Instruction mix per kernel of a gfx950 assembly listing (hipcc -S): VALU by mnemonic, with the two issue classes measured in
profiles/r01_valu_microbench.txt (VOP3-class ~4.5 clk, plain VOP1/VOP2 ~2.65 clk) priced separately."""
import collections
import re
import sys

FAST = re.compile(r"^v_(add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|mov_b32|lshrrev_b32|lshlrev_b32|ashrrev_i32|cndmask_b32|not_b32)(_e32)?$")
cur, kernels = None, collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        kernels[cur] = collections.Counter()
        continue
    if cur and re.match(r"^\s+s_endpgm", line):
        cur = None
        continue
    m = re.match(r"^\s+(v_\w+)", line)
    if cur and m:
        kernels[cur][m.group(1)] += 1
for name, c in kernels.items():
    total = sum(c.values())
    fast = sum(v for k, v in c.items() if FAST.match(k) and k.endswith("_e32"))
    clk = fast * 2.65 + (total - fast) * 4.5
    print(f"{name[:40]:40s} VALU {total:4d}  mad_u64 {c['v_mad_u64_u32']:4d}  mul_lo {c['v_mul_lo_u32']:3d}  shr_b64 {c['v_lshrrev_b64']:3d}  "
          f"VOP2-class {fast:3d}  ~issue clk per wave {clk:7.0f}")
