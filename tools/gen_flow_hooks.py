#!/usr/bin/env python3
"""Generates the deferred-epilogue hook macros of csrc/conv_flow.hip (RSR_HK_S<step>_<cell>) and splices them in place:
    python tools/gen_flow_hooks.py            # rewrites the block between the GENERATED markers
Schedule: the drain of one block (4 rows x 16 accumulator registers) is cut into 60 pieces -- per row 8 conversions of a
value pair (P), 4 transpose-scratch writes (W), 1 read-back (RB), 2 stores (ST) -- that ride behind every SECOND MFMA cell of
the next block's first 10 steps: ~2.5 VALU instructions per 32-cycle MFMA slot.  A row's stores trail its read-back by
three pieces (LDS latency), inside the next row's conversions."""
import os
import re

STEPS = 10
CELLS = 12


def pieces():
    seq = []
    pending = []  # (slot at which to emit, text)
    def emit(txt):
        seq.append(txt)
        while pending and pending[0][0] <= len(seq):
            seq.append(pending.pop(0)[1])
    for r in range(4):
        for q in range(4):
            for j in (2 * q, 2 * q + 1):
                emit("{ float v0_ = RSR_OLD[%d][0][%d], v1_ = RSR_OLD[%d][0][%d]; v0_ = __builtin_amdgcn_fmed3f(v0_, v0_ * slope, pinf); "
                     "v1_ = __builtin_amdgcn_fmed3f(v1_, v1_ * slope, pinf); pk[%d][%d] = half2v{(_Float16)v0_, (_Float16)v1_}; }"
                     % (r, 2 * j, r, 2 * j + 1, q, j & 1))
            emit("{ half4 o_ = {pk[%d][0][0], pk[%d][0][1], pk[%d][1][0], pk[%d][1][1]}; *reinterpret_cast<half4_scr*>(scr + scr_w + %d + (%d ^ scr_wx)) = o_; }"
                 % (q, q, q, q, (q >> 1) * 1024, (q & 1) << 4))
        emit("{ row_from_lds(tq, 0); }")
        at = len(seq) + 3
        pending.append((at, "{ row_store1(tq[0], od, %d, 0); }" % r))
        pending.append((at + 1, "{ row_store1(tq[1], od, %d, 1); }" % r))
    for _, txt in pending:
        seq.append(txt)
    return seq


def generate():
    seq = pieces()
    slots = STEPS * CELLS // 2
    assert len(seq) <= slots, (len(seq), slots)
    table = {}
    for s, txt in enumerate(seq):
        c = 2 * s + 1
        table[(c // CELLS, c % CELLS)] = txt
    out = []
    for st in range(STEPS):
        out.append("#define RSR_HK_S%d(c) RSR_HK_S%d_##c" % (st, st))
        for c in range(CELLS):
            body = table.get((st, c))
            if body:
                out.append("#define RSR_HK_S%d_%d __builtin_amdgcn_sched_barrier(0); %s __builtin_amdgcn_sched_barrier(0);" % (st, c, body))
            else:
                out.append("#define RSR_HK_S%d_%d __builtin_amdgcn_sched_barrier(0);" % (st, c))
    undef = []
    for st in range(STEPS):
        undef.append("#undef RSR_HK_S%d" % st)
        for c in range(CELLS):
            undef.append("#undef RSR_HK_S%d_%d" % (st, c))
    return "\n".join(out) + "\n", "\n".join(undef) + "\n"


def main():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "realsr-ncnn-vulkan_amd", "csrc", "conv_flow.hip")
    src = open(path).read()
    defs, undefs = generate()
    src, n1 = re.subn(r"(// GENERATED-HOOKS-BEGIN[^\n]*\n).*?(// GENERATED-HOOKS-END)", lambda m: m.group(1) + defs + m.group(2), src, flags=re.S)
    src, n2 = re.subn(r"(// GENERATED-UNDEFS-BEGIN[^\n]*\n).*?(// GENERATED-UNDEFS-END)", lambda m: m.group(1) + undefs + m.group(2), src, flags=re.S)
    assert n1 == 1 and n2 == 1, "markers not found"
    open(path, "w").write(src)
    print("spliced %d hook macros" % (STEPS * CELLS))


if __name__ == "__main__":
    main()
