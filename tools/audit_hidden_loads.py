"""Static check of the hand-counted `s_waitcnt vmcnt(N)` in the window epilogue (gemm_common.h): walk the gfx950 assembly of
a kernel in program order, keep the queue of outstanding vector-memory operations (they retire in issue order), and flag any
instruction that touches the destination registers of an inline-asm `global_load_dwordx4` still in the queue.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I clipa_amd/csrc -I include -ffp-contract=fast -S --cuda-device-only -o /tmp/k.s clipa_amd/csrc/gemm_nt.hip
    python tools/audit_hidden_loads.py /tmp/k.s
Also reports scratch use (a spill's loads and stores would make the counts wrong)."""
import re
import sys

text = open(sys.argv[1]).read()
kernels = re.findall(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
bad = 0
for name, body in kernels:
    if "global_load_dwordx4" not in body and "buffer_store_dwordx4" not in body:
        continue
    queue = []          # (kind, set(dest regs))
    viol = 0
    nload = 0
    in_asm = False
    for ln in body.split("\n"):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if "scratch_" in t:
            viol += 1
            print(name[:60], "SCRATCH", t)
        m = re.match(r"s_waitcnt (.*)", t)
        if m:
            v = re.search(r"vmcnt\((\d+)\)", t)
            if v:
                n = int(v.group(1))
                while len(queue) > n:
                    queue.pop(0)
            continue
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
            regs.update(range(int(a), int(b) + 1))
        for a in re.findall(r"\bv(\d+)\b", t):
            regs.add(int(a))
        for kind, dest in queue:
            if kind == "hidden" and dest & regs:
                viol += 1
                print(name[:60], "TOUCHES PENDING LOAD:", t)
        if t.startswith("global_load_dwordx4") and in_asm:
            a, b = re.search(r"v\[(\d+):(\d+)\]", t).groups()
            queue.append(("hidden", set(range(int(a), int(b) + 1))))
            nload += 1
        elif re.match(r"(buffer|global|flat)_(load|store|atomic)", t):
            queue.append(("other", set()))
    # the wait in front of the next tile's first K step leaves exactly the epilogue's stores in flight: 16 per wave, 32 with the
    # pre-activation copy; the epilogue may be emitted more than once (loop rotation), never partially
    after = re.search(r"s_waitcnt vmcnt\((16|32)\) lgkmcnt\(0\)", body)
    nstore = len(re.findall(r"^\s*buffer_store_dwordx4", body, re.M))
    if after and (nstore == 0 or nstore % int(after.group(1)) != 0):
        viol += 1
        print(name[:60], f"STORE COUNT {nstore} is not a multiple of the counted {after.group(1)}")
    print(f"{name[:90]}: {nload} hidden loads, {viol} violations")
    bad += viol
sys.exit(1 if bad else 0)
