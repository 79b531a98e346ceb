#!/usr/bin/env python3
"""Static instruction mix per basic block of a gfx950 .s file (hipcc -S --cuda-device-only).
usage: isa_blocks.py file.s [kernel-substring] [min_instrs]"""
import re, sys
path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""; minn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cur_k = None; blocks = []; blk = None
for ln, line in enumerate(open(path), 1):
    s = line.strip()
    m = re.match(r"^([A-Za-z_][\w$.]*):", s)
    if m and not s.startswith(".L"):
        cur_k = m.group(1); blk = None; continue
    if cur_k is None or want not in cur_k: continue
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", s)
    if m or blk is None:
        blk = {"name": m.group(1) if m else "entry", "line": ln, "note": (m.group(2).strip() if m else ""), "v": 0, "s": 0, "lds": 0, "vmem": 0, "smem": 0, "wait": 0, "bar": 0, "br": 0}
        blocks.append(blk)
        if m: continue
    op = s.split()[0] if s and not s.startswith((";", ".")) else None
    if not op: continue
    if op.startswith("v_"): blk["v"] += 1
    elif op.startswith("ds_"): blk["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): blk["vmem"] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer"): blk["smem"] += 1
    elif op.startswith("s_waitcnt"): blk["wait"] += 1
    elif op.startswith("s_barrier"): blk["bar"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): blk["br"] += 1
    elif op.startswith("s_"): blk["s"] += 1
for b in blocks:
    tot = b["v"] + b["s"] + b["lds"] + b["vmem"]
    if tot >= minn:
        print(f'{b["line"]:6d} {b["name"]:12s} V={b["v"]:4d} S={b["s"]:4d} LDS={b["lds"]:3d} VM={b["vmem"]:3d} SM={b["smem"]:2d} wait={b["wait"]:3d} bar={b["bar"]} br={b["br"]}  {b["note"][:60]}')
