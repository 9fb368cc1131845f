#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel of libdsm_mi355x.so, from the device assembly (no GPU needed):
VGPRs USED (NumVgprs + NumAgprs) against VGPRs ALLOCATED (NumVGPRsForWavesPerEU, what .amdhsa_next_free_vgpr reserves per wave),
static LDS, scratch (spills), occupancy as the compiler sees it.  A static LDS array that bounds the occupancy makes the compiler
pad the allocation up to that occupancy (r04: 136 allocated for 88 used) — registers other streams' workgroups cannot use beside the
kernel; such rows are flagged PADDED.  usage: tools/kernel_resources.py [substring ...]   (takes about 80 s: one device compile)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "delayed-streams-modeling_amd", "csrc", "dsm_engine.hip")
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "dsm.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", "-o", asm, SRC], check=True, stderr=subprocess.DEVNULL)
    s = open(asm).read()
rows = []
for m in re.finditer(r"; Kernel info:\n(.*?); Occupancy: (\d+)", s, re.S):
    blk = m.group(1)
    get = lambda k: int(re.search(r"; %s: (\d+)" % k, blk).group(1))
    pre = s[max(0, m.start() - 800):m.start()]
    names = re.findall(r"\.set (\S+)\.has_indirect_call", pre)
    if not names:
        continue
    name = subprocess.run(["c++filt", names[-1]], capture_output=True, text=True).stdout.strip()
    name = name.replace("unsigned short", "bf16").replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    rows.append((name, get("TotalNumVgprs"), get("NumVGPRsForWavesPerEU"), get("LDSByteSize"), get("ScratchSize"), int(m.group(2))))
subs = sys.argv[1:]
print(f"{'kernel':64s} {'used':>5s} {'alloc':>5s} {'LDS':>6s} {'scr':>4s} occ")
for name, used, alloc, lds, scr, occ in sorted(rows, key=lambda r: (-r[2], r[0])):
    if subs and not any(x in name for x in subs):
        continue
    flag = "  PADDED" if alloc - used >= 16 else ("  SPILLS" if scr else "")
    print(f"{name[:64]:64s} {used:5d} {alloc:5d} {lds:6d} {scr:4d} {occ:3d}{flag}")
