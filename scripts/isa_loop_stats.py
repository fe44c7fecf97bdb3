#!/usr/bin/env python3
"""Static view of k_raytrace's walk loop (no GPU needed): compile k_raytrace.hip to gfx950 assembly and, per kernel build, find the
innermost loop that requests walk records (the basic blocks between the loop header and its back edge), count its instructions by
class and list any scratch access inside it.  The kernel is issue bound, so instructions per step x steps per ray
is the first-order model of a change's effect before it is measured on the GPU.

    python scripts/isa_loop_stats.py [filter substring of the mangled name]
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.environ.get("ISA_SRC") or os.path.join(ROOT, "renderer_amd", "csrc", "k_raytrace.hip")
FLAGS = ("-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function -Wno-unused-variable --cuda-device-only -S " + os.environ.get("ISA_EXTRA", "")).split()

def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    out = os.path.join(tempfile.gettempdir(), "k_raytrace_isa.s")
    if not (len(sys.argv) > 2 and sys.argv[2] == "reuse" and os.path.exists(out)):
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    meta = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(
        r'\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.|\n)*?\.vgpr_count:\s+(\d+)', txt)}
    funcs = re.split(r'\n(?=_Z10k_raytrace\S+:)', txt)
    for f in funcs:
        m = re.match(r'(_Z10k_raytrace\S+?):', f)
        if not m or flt not in m.group(1): continue
        name = m.group(1)
        body = f[:f.index("s_endpgm")] if "s_endpgm" in f else f
        lines = body.split("\n")
        # basic blocks carry LLVM's loop annotations: "=>This [Inner] Loop Header: Depth=d", "in Loop: Header=BBx_y Depth=d",
        # "Parent Loop BBx_y Depth=d" -- blocks the compiler moved behind the back edge (the rare exact tests) still name their loop
        blocks, cur, parent = [], None, {}
        for l in lines:
            t = l.strip()
            mlab = re.match(r'\.LBB(\d+_\d+):', t)
            if mlab or t.startswith("; %bb."):
                cur = {"label": "BB" + mlab.group(1) if mlab else None, "loops": [], "ins": [], "hdr": False, "cold": False}
                blocks.append(cur)
                if "Loop Header" in t and mlab: cur["loops"].append(cur["label"]); cur["hdr"] = True
                mh = re.search(r'in Loop: Header=(BB\d+_\d+)', t)
                if mh: cur["loops"].append(mh.group(1))
                mp = re.search(r'Parent Loop (BB\d+_\d+)', t)
                if mp and mlab: cur["par"] = mp.group(1)
                continue
            if cur is None: continue
            if t.startswith(";") and not cur["ins"]:
                mp = re.search(r'Parent Loop (BB\d+_\d+)', t)
                if mp and "par" not in cur: cur["par"] = mp.group(1)
                if "Loop Header" in t and cur["label"]:
                    cur["hdr"] = True
                    if cur["label"] not in cur["loops"]: cur["loops"].insert(0, cur["label"])
                if cur["hdr"] and "par" in cur and cur["label"] not in parent: parent[cur["label"]] = cur["par"]
                continue
            if t and not t.startswith((";", ".")) and not t.endswith(":"): cur["ins"].append(t)
        def ancestors(h):
            out = [h]
            while out[-1] in parent: out.append(parent[out[-1]])
            return out
        members = {}
        for b in blocks:
            for h in (ancestors(b["loops"][0]) if b["loops"] else []): members.setdefault(h, []).append(b)
        best = None
        size = lambda h: sum(len(b["ins"]) for b in members[h])
        for h, bl in members.items():
            ins_h = [x for b in bl for x in b["ins"]]
            # (the walk loop: it requests wide records AND holds the work sharing's hand-over -- the ray through ~20 ds_bpermute; the loop
            #  that tests the queued leaves also loads records and shuffles a ray, nine values of it)
            walkish = sum("ds_bpermute" in x for x in ins_h) >= 15 or not any("ds_bpermute" in x for b2 in blocks for x in b2["ins"])
            if sum("global_load_dwordx4" in x for x in ins_h) >= 4 and any("ds_write" in x for x in ins_h) and walkish:
                if best is None or size(h) < size(best): best = h
        if not best: continue
        # blocks the compiler laid out behind the loop's back edge are its cold paths (the exact box tests)
        seen_back = False
        for b in blocks:
            if b in members[best] and seen_back: b["cold"] = True
            if any(re.match(r's_branch\s+\.L' + best + r'$', x) for x in b["ins"]) and b in members[best] and not seen_back: seen_back = True
        ins = [x for b in members[best] if not b["cold"] for x in b["ins"]]
        cold = sum(len(b["ins"]) for b in members[best] if b["cold"])
        cls = lambda p: sum(1 for x in ins if x.startswith(p))
        valu = sum(1 for x in ins if x.startswith("v_"))
        print("%-72s vgpr %3d scratch %3d | loop %4d (+%d cold) instr: valu %3d (pk %2d) salu %3d vmem %2d ds %2d branch %2d scratch-in-loop %d" % (
            name[:72], meta.get(name, (0, 0))[1], meta.get(name, (0, 0))[0], len(ins), cold, valu, cls("v_pk_"), cls("s_") - cls("s_cbranch") - cls("s_branch") - cls("s_waitcnt") - cls("s_nop"),
            cls("global_"), cls("ds_"), cls("s_cbranch") + cls("s_branch"), cls("scratch_")))

if __name__ == "__main__":
    main()
