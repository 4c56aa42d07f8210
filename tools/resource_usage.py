#!/usr/bin/env python3
"""Cuts hipcc's kernel-resource-usage remarks and the ISA of csrc/sri_apply.hip (`make -C <pkg>/csrc asm` -> build/asm/)
down to one line per shipped variant of sri::apply_kernel: registers, spills, occupancy, and how many v_readlane /
v_writelane (scalar-register spill traffic) and other vector-ALU instructions sit in basic blocks that hold MFMAs.

    python tools/resource_usage.py [profiles/rNN_resource_usage.txt]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, "build/asm/resource_usage.txt")).read()
isa = open(os.path.join(ROOT, "build/asm/sri_apply-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
GEO = {"1,4,4,1,3,0": "flat 64x64   Geo<1,4,4,1,3>", "2,2,4,4,2,0": "mid  128x128 Geo<2,2,4,4,2>", "4,2,4,4,3,0": "tall 256x128 Geo<4,2,4,4,3>",
       "1,4,4,4,2,0": "wide 64x256  Geo<1,4,4,4,2>", "1,4,2,4,2,0": "wide 32x256  Geo<1,4,2,4,2>", "4,2,4,4,3,1": "strip 272x128 Geo<4,2,4,4,3,1>", "4,2,4,4,3,2": "mini 256+4x128 Geo<4,2,4,4,3,2>", "2,2,4,4,2,2": "mini 128+4x128 Geo<2,2,4,4,2,2>", "1,4,4,1,2,2": "mini 64+4x64   Geo<1,4,4,1,2,2>",
       "1,4,6,2,2,0": "g96  96x128  Geo<1,4,6,2,2>", "1,4,10,1,2,0": "g160 160x64  Geo<1,4,10,1,2>", "2,2,6,2,2,0": "g192 192x64  Geo<2,2,6,2,2>",
       "1,4,14,1,2,0": "g224 224x64  Geo<1,4,14,1,2>"}
SCR, OCC = r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]"
out = ["# Register / spill / occupancy report of every shipped variant of sri::apply_kernel (csrc/sri_apply.hip).",
       "# `make -C <pkg>/csrc asm` (hipcc -Rpass-analysis=kernel-resource-usage, --offload-arch=gfx950) + tools/resource_usage.py.",
       "# lane/VALU: v_readlane + v_writelane (scalar-register spill traffic) and all other vector-ALU instructions in the basic",
       "# blocks (label to branch) of the ISA that hold at least one v_mfma (stage bodies AND tile ends; the steady stage bodies are the blocks with",
       "# 28-64 MFMAs and no global_store).  EPI 1 = fused quaternion step, ASC = ascending k order, LEAN = the twin of a TAIL variant",
       "# without the edge handling of the panel's last column (launched where run() finds the bytes past the panel readable).",
       f"{'geometry':28s} DMA R1 TAIL EPI ASC LEAN | VGPR SGPR Sspill Vspill scratch occ | lane / VALU in MFMA blocks | lane ops in store-free MFMA blocks"]
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split(" ")[0]
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    mg = re.search(r"GeoILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEE", name)
    m = re.search(r"GeoILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi\d+EEELb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d+)ELi(\d)ELb(\d)ELb(\d)", name)
    if mg is None or m is None:
        continue
    i = isa.index(name + ":")
    body = isa[i:isa.index(".Lfunc_end", i)]
    lane = valu = steady_lane = 0
    cur = []

    def flush(cur):
        global lane, valu, steady_lane
        if any(x.startswith("v_mfma") for x in cur):
            l = sum(1 for x in cur if x.startswith(("v_readlane", "v_writelane")))
            lane += l
            valu += sum(1 for x in cur if x.startswith("v_") and not x.startswith(("v_mfma", "v_readlane", "v_writelane")))
            if not any(x.startswith("global_store") for x in cur):
                steady_lane += l

    # basic blocks in the strict sense: a label starts one, a branch ends one (round 5: instructions behind a conditional branch
    # that skips them in every stage but a tile's first — issue_qe, issue_setup — used to be counted with the MFMAs in front of it)
    for ln in body.split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", ln):
            flush(cur)
            cur = []
        else:
            t = ln.strip()
            if t and not t.startswith((";", ".")):
                cur.append(t)
                if t.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
                    flush(cur)
                    cur = []
    flush(cur)
    out.append(f"{GEO.get(','.join(mg.groups()), ','.join(mg.groups())):28s} {m.group(6):>3s} {m.group(7):>2s} {m.group(8):>4s} {m.group(11):>3s} {m.group(12):>3s} {m.group(13):>4s} | "
               f"{g('VGPRs'):>4s} {g('TotalSGPRs'):>4s} {g('SGPRs Spill'):>6s} {g('VGPRs Spill'):>6s} {g(SCR):>7s} {g(OCC):>3s} | "
               f"{lane:4d} / {valu:4d} | {steady_lane}")
text = "\n".join(out) + "\n"
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text)
print(text)
