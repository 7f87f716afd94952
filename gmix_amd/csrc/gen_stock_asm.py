#!/usr/bin/env python3
"""Generates gmx_stock_asm.inc: the instruction streams of gmx_stock.hip's hot phases.

Why generated assembly: a stream's bank of the stock topology is 174 MiB, so a full HBM holds
about as many streams as the chip has SIMDs -- one wave per SIMD, and every instruction that
wave issues, scalar or vector, is on the critical path.  hipcc's code for this arithmetic needs
~2400 instructions per bit (selects with recomputed lane masks, SGPR spills through
v_writelane, VGPR<->AGPR shuffling); the streams below need ~900 and perform the same IEEE
operations in the same order (mul, then add -- never fused).

Register map.  The kernels are compiled with amdgpu_num_vgpr(96) / amdgpu_num_sgpr(...) so
that hipcc itself only allocates v0..v43, a0..a43 and the low SGPRs; everything above is
reserved for the streams below and referenced by name (tests/test_abi.py checks that no
compiler-generated instruction touches the reserved ranges):
  v44..v47                       temporaries of the streams
  X(j)  = v[48+j],  j = 0..91    the bit's inputs (broadcast-read from LDS)
  W(j)  = v[140+j], j = 0..115   the lane's resident row (lane m = mixer m)
  a[140..255]                    the row prefetched lane-private (the 9 quads of a layer-1 / final lane always;
                                 all 29 quads of a layer-0 lane in launches that do not stage rows through LDS)
  a[44..175]                     33 quads on their way from the write-back image to the stores (staged launches)
  O0(i) = s[64+i],  i = 0..23    layer-0 outputs;  O1(i) = s[88+i], i = 0..7  layer-1 outputs

Hazards honoured by construction (gfx940/gfx950; the spacing hipcc itself keeps):
  VALU writes a VGPR -> v_readlane of it: >= 1 wait state;
  v_readlane writes an SGPR -> VALU reads it: >= 2 wait states.
"""
import os

TB, XB, WB, AB, O0, O1 = 44, 48, 140, 140, 70, 94
N, L0, L1, M = 90, 24, 8, 33
NQX = (N + 3) // 4   # 23 quads of inputs
NQW = 29             # quads of a layer-0 row (113 weights)
NQA = 9              # quads of a layer-1 / final row that hold weights (33 of the 64 floats stored)
RESERVED = {"v": (44, 255), "a": (44, 255), "s": (70, 101)}


def W(j):
    return f"v{WB + j}"


def W2(j):
    assert j % 2 == 0
    return f"v[{WB + j}:{WB + j + 1}]"


def W4(q):
    return f"v[{WB + 4 * q}:{WB + 4 * q + 3}]"


def X(j):
    return f"v{XB + j}"


def X2(j):
    return f"v[{XB + j}:{XB + j + 1}]"


def X4(q):
    return f"v[{XB + 4 * q}:{XB + 4 * q + 3}]"


def A4(q):
    return f"a[{AB + 4 * q}:{AB + 4 * q + 3}]"


LLVM_MC = "/opt/rocm/lib/llvm/bin/llvm-mc"
_DUMMY = {  # operand placeholders -> registers of the right class, for sizing only
    "sv": "s[10:11]", "mask": "s[12:13]", "seenlo": "s14", "seenhi": "s15", "nm": "s[14:15]", "em": "s[14:15]", "m": "s[14:15]", "img": "s16",
    "sm0": "s17", "p": "v[2:3]", "p0": "v[4:5]", "p1": "v[6:7]", "up2": "v[8:9]", "sc2": "v[8:9]",
}


def _sizes(lines):
    """Encoded size of every instruction (4 or 8 bytes on gfx950), from the assembler itself."""
    import re
    import subprocess
    text = []
    for l in lines:
        text.append(re.sub(r"%\[(\w+)\]", lambda m: _DUMMY.get(m.group(1), "v1"), l))
    out = subprocess.run([LLVM_MC, "-arch=amdgcn", "-mcpu=gfx950", "-show-encoding"], input="\n".join(text) + "\n",
                         capture_output=True, text=True, check=True).stdout
    sz = [len(m.group(1).split(",")) for m in re.finditer(r"encoding: \[([^\]]*)\]", out)]
    assert len(sz) == len(lines), (len(sz), len(lines))
    return sz


_WIDENABLE = ("v_add_f32 ", "v_sub_f32 ", "v_mul_f32 ", "v_mov_b32 ", "v_cndmask_b32 ")


def align8(lines):
    """Every 8-byte instruction on an 8-byte boundary.  A wave that runs alone on its SIMD pays for an
    8-byte instruction that straddles a boundary (measured here: the update stream, all packed ops, took
    12 % longer when the block happened to start at 4 mod 8 -- MI355X_MICROARCH.md notes the same for
    hand-written streams).  No instruction is added where one can be re-encoded: the nearest preceding
    4-byte VALU instruction takes its 8-byte VOP3 form (same operation); failing that an s_nop goes in."""
    sz = _sizes(lines)
    out, osz, off = [], [], 0
    for ins, n in zip(lines, sz):
        if n == 8 and off % 8 == 4:
            k = len(out) - 1
            while k >= 0 and osz[k] == 4 and not out[k].startswith(_WIDENABLE):
                k -= 1
            if k >= 0 and osz[k] == 4:
                mnem, rest = out[k].split(" ", 1)
                out[k], osz[k] = f"{mnem}_e64 {rest}", 8
            else:
                out.append("s_nop 0")
                osz.append(4)
            off += 4
        out.append(ins)
        osz.append(n)
        off += n
    assert _sizes(out) == osz
    return [".p2align 3"] + out


def emit(name, lines):
    lines = align8(lines)
    body = "".join(f'  "{l}\\n\\t" \\\n' for l in lines)
    return f"#define {name} \\\n{body}  \"\"\n\n"


def under_mask(body):
    return ["s_mov_b64 %[sv], exec", "s_mov_b64 exec, %[mask]"] + body + ["s_mov_b64 exec, %[sv]"]


def loads(first, last):
    """lane-private row pieces HBM -> AGPRs (the small rows of the layer-1 / final lanes: 9 pieces
    of 16 bytes in at most 9 lanes -- too little to be worth staging)"""
    return under_mask([f"global_load_dwordx4 {A4(q)}, %[p], off offset:{16 * q}" for q in range(first, last)])


def stores(first, last):
    return under_mask([f"global_store_dwordx4 %[p], {W4(q)}, off offset:{16 * q}" for q in range(first, last)])


def adopt(first, last):
    return under_mask([f"v_accvgpr_read_b32 {W(j)}, a{AB + j}" for j in range(4 * first, 4 * last)])


SP = 70        # s70..s77: four rotating SGPR pairs for row addresses (O0/O1 are dead outside forward..update)
STG = 44       # a44..a175: 33 quads, the write-back's way from the image to the stores
PITCH = 528    # bytes between rows of a staging image: 33 quads, odd, so transposed accesses do not conflict


def _row_addr(r, lo, hi):
    p = SP + 2 * (r % 4)
    return [f"v_readlane_b32 s{p}, {lo}, {r}", f"v_readlane_b32 s{p + 1}, {hi}, {r}"]


def _row_lanes(r):
    """a layer-0 row is 512 bytes = 32 lanes x 16, a layer-1 / final row 256 bytes = 16 lanes"""
    return "-1" if r < L0 else "0xffff"


def fetch_rows(NR=M):
    """The rows of the mixers in %[nm] (64-bit), HBM -> staging image, one coalesced LDS-DMA per row (32 or
    16 lanes x 16 contiguous bytes) instead of 29 lane-private 16-byte loads that touch as many different
    lines as there are lanes.  The row address lives in lane r (%[plo]/%[phi]) and is made scalar two rows
    ahead of its use (v_readlane -> VMEM address: 5 wait states); an unchanged row issues with exec = 0.
    This is the stream for bits where most rows move (byte boundaries); gmx_stock.hip walks the set bits
    of %[nm] itself when few do."""
    l = ["s_mov_b64 %[sv], exec", "s_mov_b32 %[sm0], m0", "s_mov_b32 exec_hi, 0"]
    l += _row_addr(0, "%[plo]", "%[phi]") + _row_addr(1, "%[plo]", "%[phi]")
    for r in range(NR):
        if r + 2 < NR:
            l += _row_addr(r + 2, "%[plo]", "%[phi]")
        p = SP + 2 * (r % 4)
        l.append(f"s_bitcmp1_b64 %[nm], {r}")
        l.append(f"s_cselect_b32 exec_lo, {_row_lanes(r)}, 0")
        l.append(f"s_add_u32 m0, %[img], {PITCH * r}")
        l.append("s_nop 0")
        l.append(f"global_load_lds_dwordx4 %[voff], s[{p}:{p + 1}]")
    l += ["s_mov_b32 m0, %[sm0]", "s_mov_b64 exec, %[sv]"]
    return l


def fetch_sparse():
    """The same for a few rows: a loop over the set bits of %[m] (64-bit, consumed), 11 instructions a row.
    s70..s74 are free here (O0 is dead outside forward..update).  The six scalar instructions between the
    second v_readlane and the LDS-DMA are its wait states (v_readlane -> VMEM address: 5; m0 -> LDS-DMA: 1)."""
    return [f"s_mov_b32 s{SP + 3}, 0xffff", "s_mov_b64 %[sv], exec", "s_mov_b32 %[sm0], m0", "s_mov_b32 exec_hi, 0",
            f"s_movk_i32 s{SP + 4}, {PITCH}", "s_nop 0",  # the loop starts at 4 mod 8: nothing in it straddles
            f"1: s_ff1_i32_b64 s{SP + 2}, %[m]",
            f"v_readlane_b32 s{SP}, %[plo], s{SP + 2}",
            f"v_readlane_b32 s{SP + 1}, %[phi], s{SP + 2}",
            f"s_bitset0_b64 %[m], s{SP + 2}",
            f"s_cmp_lt_u32 s{SP + 2}, {L0}",
            f"s_cselect_b32 exec_lo, -1, s{SP + 3}",
            f"s_mul_i32 s{SP + 2}, s{SP + 2}, s{SP + 4}",
            f"s_add_u32 m0, %[img], s{SP + 2}",
            "s_cmp_lg_u64 %[m], 0",
            f"global_load_lds_dwordx4 %[voff], s[{SP}:{SP + 1}]",
            "s_cbranch_scc1 1b",
            "s_mov_b32 m0, %[sm0]", "s_mov_b64 exec, %[sv]"]


def evict_rows(NR=M):
    """... and back: the rows the lanes of %[em] wrote into the write-back image (to_image) leave as one
    coalesced store per row.  All 33 rows of the image are read (into a44..a175: no scalar work, one
    wait); a row that is not being replaced stores with exec = 0."""
    l = ["s_mov_b64 %[sv], exec"]
    for r in range(NR):
        l.append(f"ds_read_b128 a[{STG + 4 * r}:{STG + 4 * r + 3}], %[va] offset:{PITCH * r}")
    l.append("s_mov_b32 exec_hi, 0")
    l += _row_addr(0, "%[plo]", "%[phi]") + _row_addr(1, "%[plo]", "%[phi]")
    for r in range(NR):
        if r + 2 < NR:
            l += _row_addr(r + 2, "%[plo]", "%[phi]")
        p = SP + 2 * (r % 4)
        l.append(f"s_bitcmp1_b64 %[em], {r}")
        l.append(f"s_cselect_b32 exec_lo, {_row_lanes(r)}, 0")
        if r == 0:
            l.append("s_waitcnt lgkmcnt(0)")
        l.append(f"global_store_dwordx4 %[voff], a[{STG + 4 * r}:{STG + 4 * r + 3}], s[{p}:{p + 1}]")
    l.append("s_mov_b64 exec, %[sv]")
    return l


def to_image(first, last):
    """the resident row of the lanes in the mask -> the lane's row of the write-back image in LDS
    (%[l] = the lane's row address: rows are 528 bytes apart, so the 16 lanes of a ds group hit 16
    different bank quads)"""
    return under_mask([f"ds_write_b128 %[l], {W4(q)} offset:{16 * q}" for q in range(first, last)])


def from_image(first, last):
    """the prefetched row of the lanes in the mask: staging image in LDS -> the resident row"""
    return under_mask([f"ds_read_b128 {W4(q)}, %[l] offset:{16 * q}" for q in range(first, last)])


def load_x(l, consume):
    """Broadcast-read the 23 input quads from LDS into X; consume(q) emits the work that needs quad q.
    Every instruction costs a lone wave its 4-cycle issue slot, an s_waitcnt that is already satisfied
    included (scripts/ubench/issue_ubench.hip), so the reads are waited for four quads at a time: 12 reads
    go out at once (lgkmcnt is a 4-bit counter: never more than 15 in flight), then each group of four
    costs one wait and is followed by the next four reads."""
    group, ahead = 4, 12
    l.append("s_waitcnt lgkmcnt(0)")
    issued = min(ahead, NQX)
    for q in range(issued):
        l.append(f"ds_read_b128 {X4(q)}, %[xaddr] offset:{16 * q}")
    for g0 in range(0, NQX, group):
        g1 = min(NQX, g0 + group)
        l.append(f"s_waitcnt lgkmcnt({issued - g1})")
        more = min(NQX, issued + group)
        for q in range(issued, more):
            l.append(f"ds_read_b128 {X4(q)}, %[xaddr] offset:{16 * q}")
        issued = more
        for q in range(g0, g1):
            consume(q)


def chain_l0(l, with_loads=True, from_zero=False):
    """layer 0, inputs 0..89 (mixer.cpp:56-59): acc = acc + x*w, left to right (from_zero: the first sum is
    0.0f + x*w with the constant as an operand, acc need not be cleared first)"""
    def consume(q):
        # products two at a time (same IEEE multiply), the sum strictly one after the other
        for h in range(2):
            j = 4 * q + 2 * h
            if j < N:
                t = TB + 2 * ((j // 2) % 2)
                l.append(f"v_pk_mul_f32 v[{t}:{t + 1}], {X2(j)}, {W2(j)}")
                l.append(f"v_add_f32 %[acc], {'0' if from_zero and j == 0 else '%[acc]'}, v{t}")
                if j + 1 < N:
                    l.append(f"v_add_f32 %[acc], %[acc], v{t + 1}")
    if with_loads:
        load_x(l, consume)
    else:
        for q in range(NQX):
            consume(q)


def forward():
    """33 x Mixer::Predict (mixer.cpp:51-106), every lane its own mixer.  A lane that takes no
    part in a step sees a zero weight there (the stored padding beyond weight_size; a row never
    learned is all zeros), so `acc + o*0` leaves it alone -- exact as long as every value is
    finite, which the kernel checks afterwards (else it redoes the bit with forward_exact)."""
    l = []
    chain_l0(l, from_zero=True)
    # layer-0 cascade (mixer.cpp:60-64) merged with the sums of layer 1 / final over the
    # layer-0 outputs (mixer.cpp:66-68, 82-84): O0(i) is read once and feeds both.  The hazard slots
    # (VALU -> v_readlane of its result: 1 wait state; v_readlane -> VALU reading the SGPR: 2) hold
    # useful instructions where there are any -- a nop costs a lone wave a full issue slot per wait state.
    for i in range(L0):
        if i >= 1:
            l.append(f"v_mul_f32 %[t1], s{O0 + i - 1}, {W(i - 1)}")
        else:
            l.append("s_mov_b32 vcc_hi, 0")  # (for the layer-1 selects below)
        l.append(f"v_readlane_b32 s{O0 + i}, %[acc], {i}")
        if i >= 1:
            l.append("v_add_f32 %[a1], %[a1], %[t1]")
        else:
            l.append("v_mov_b32 %[a1], 0")
        l.append("s_nop 0")
        if i < L0 - 1:
            l.append(f"v_mul_f32 %[t0], s{O0 + i}, {W(N + i)}")
            l.append("v_add_f32 %[acc], %[acc], %[t0]")
    l.append(f"v_mul_f32 %[t1], s{O0 + L0 - 1}, {W(L0 - 1)}")
    l.append("v_add_f32 %[a1], %[a1], %[t1]")
    # From here on %[a1] is the running sum of lanes 24..32 (garbage in the layer-0 lanes, whose
    # outputs stay in %[acc]).  Layer-1 cascade, each mixer's skip input closing its chain
    # (mixer.cpp:69-80); lane 32, the final mixer, collects the layer-1 outputs on the way
    # (mixer.cpp:85-90).  The skip product of the NEXT mixer (the final mixer's after the last,
    # mixer.cpp:91-97; weight 32 of the layer-1 lanes is padding) fills one of the two wait states
    # behind each v_readlane; it goes to the chain's product registers, which are free by now.
    sk = lambda i: f"v{TB + (i % 2)}"
    l.append(f"v_mul_f32 {sk(0)}, %[vskip], {W(L0)}")
    for i in range(L1):
        l.append(f"v_add_f32 %[t1], %[a1], {sk(i)}")
        l.append(f"s_mov_b32 vcc_lo, {hex(1 << (L0 + i))}")
        l.append(f"v_readlane_b32 s{O1 + i}, %[t1], {L0 + i}")
        l.append(f"v_mul_f32 {sk(i + 1)}, %[vskip], {W(L0 + i + 1)}")
        l.append("s_nop 0")
        l.append(f"v_mul_f32 %[t0], s{O1 + i}, {W(L0 + i)}")
        l.append("v_add_f32 %[t0], %[a1], %[t0]")
        l.append("v_cndmask_b32 %[a1], %[t0], %[t1], vcc")
    l.append(f"v_add_f32 %[a1], %[a1], {sk(L1)}")
    l.append("s_mov_b32 vcc_lo, 0xffffff")
    l.append("v_cndmask_b32 %[acc], %[a1], %[acc], vcc")
    return l


def forward_exact():
    """The same chains with every step under the exec mask of the lanes the reference visits
    (and only rows that exist, mixer.cpp:52-55): no value reaches a lane it does not belong to,
    whatever it is.  Slower (more scalar instructions); used when forward() met a non-finite."""
    l = ["v_mov_b32 %[acc], 0", "v_mov_b32 %[a1], 0"]
    load_x(l, lambda q: None)  # X in every lane: the update reads it in all layer-0 lanes
    l.append("s_mov_b64 %[sv], exec")
    l.append("s_and_b32 exec_lo, %[seenlo], 0xffffff")
    l.append("s_mov_b32 exec_hi, 0")
    chain_l0(l, with_loads=False)
    for i in range(L0):
        l.append("s_nop 0")
        l.append(f"v_readlane_b32 s{O0 + i}, %[acc], {i}")
        if i < L0 - 1:
            l.append(f"s_and_b32 exec_lo, %[seenlo], {hex(0xffffff & ~((2 << i) - 1))}")
            l.append("s_nop 0")
            l.append(f"v_mul_f32 %[t0], s{O0 + i}, {W(N + i)}")
            l.append("v_add_f32 %[acc], %[acc], %[t0]")
    l.append("s_and_b32 exec_lo, %[seenlo], 0xff000000")
    l.append("s_and_b32 exec_hi, %[seenhi], 1")
    for i in range(L0):
        t = "%[t0]" if i % 2 == 0 else "%[t1]"
        l.append(f"v_mul_f32 {t}, s{O0 + i}, {W(i)}")
        l.append(f"v_add_f32 %[a1], %[a1], {t}")
    for i in range(L1):
        l.append(f"s_and_b32 exec_lo, %[seenlo], {hex(1 << (L0 + i))}")
        l.append("s_mov_b32 exec_hi, 0")
        l.append(f"v_mul_f32 %[t0], %[vskip], {W(L0 + i)}")
        l.append("v_add_f32 %[a1], %[a1], %[t0]")
        l.append("s_nop 0")
        l.append(f"v_readlane_b32 s{O1 + i}, %[a1], {L0 + i}")
        l.append(f"s_and_b32 exec_lo, %[seenlo], {hex(0xff000000 & ~((2 << (L0 + i)) - 1))}")
        l.append("s_and_b32 exec_hi, %[seenhi], 1")
        l.append(f"v_mul_f32 %[t0], s{O1 + i}, {W(L0 + i)}")
        l.append("v_add_f32 %[a1], %[a1], %[t0]")
    l.append("s_mov_b32 exec_lo, 0")
    l.append("s_and_b32 exec_hi, %[seenhi], 1")
    l.append(f"v_mul_f32 %[t0], %[vskip], {W(L0 + L1)}")
    l.append("v_add_f32 %[a1], %[a1], %[t0]")
    l.append("s_mov_b64 exec, %[sv]")
    l.append("s_mov_b32 vcc_lo, 0xffffff")
    l.append("s_mov_b32 vcc_hi, 0")
    l.append("v_cndmask_b32 %[acc], %[a1], %[acc], vcc")
    return l


def load_x_only():
    l = []
    load_x(l, lambda q: None)
    return l


def outputs_to_sgprs():
    """O0/O1 from a per-lane vector of mixer outputs (learn-only launches of the per-bit API)."""
    l = ["s_nop 0"]
    for i in range(L0):
        l.append(f"v_readlane_b32 s{O0 + i}, %[acc], {i}")
    for i in range(L1):
        l.append(f"v_readlane_b32 s{O1 + i}, %[acc], {L0 + i}")
    l.append("s_nop 1")
    return l


def update():
    """The weight sweeps of 33 x Mixer::Learn (mixer.cpp:129-172): w -= update * x (mul, then
    subtract), lanes and elements selected with exec instead of per-element selects."""
    l = ["s_mov_b64 %[sv], exec"]
    # layer-0 lanes, inputs 0..89: two elements per instruction
    l.append("s_mov_b64 exec, 0xffffff")
    for j in range(0, N, 2):
        t = "%[p0]" if (j // 2) % 2 == 0 else "%[p1]"
        l.append(f"v_pk_mul_f32 {t}, {X2(j)}, %[up2]")
        l.append(f"v_pk_add_f32 {W2(j)}, {W2(j)}, {t} neg_lo:[0,1] neg_hi:[0,1]")
    # layer-0 lanes, weight 90+i multiplies the output of mixer i, lanes i+1..23 only
    for i in range(L0 - 1):
        t = "%[t0]" if i % 2 == 0 else "%[t1]"
        l.append(f"s_bitset0_b32 exec_lo, {i}")
        l.append(f"v_mul_f32 {t}, s{O0 + i}, %[upd]")
        l.append(f"v_sub_f32 {W(N + i)}, {W(N + i)}, {t}")
    # layer-1 and final lanes: the 24 layer-0 outputs
    l.append("s_mov_b32 exec_lo, 0xff000000")
    l.append("s_mov_b32 exec_hi, 1")
    l.append("v_mul_f32 %[ts], %[vskip], %[upd]")
    for j in range(0, L0, 2):
        t = "%[p0]" if (j // 2) % 2 == 0 else "%[p1]"
        l.append(f"v_pk_mul_f32 {t}, s[{O0 + j}:{O0 + j + 1}], %[up2]")
        l.append(f"v_pk_add_f32 {W2(j)}, {W2(j)}, {t} neg_lo:[0,1] neg_hi:[0,1]")
    # weight 24+i multiplies layer-1 output i, lanes 24+i+1..32 only
    for i in range(L1):
        t = "%[t0]" if i % 2 == 0 else "%[t1]"
        l.append(f"s_bitset0_b32 exec_lo, {L0 + i}")
        l.append(f"v_mul_f32 {t}, s{O1 + i}, %[upd]")
        l.append(f"v_sub_f32 {W(L0 + i)}, {W(L0 + i)}, {t}")
    # each mixer's own skip weight: lane 24+k holds it at index 24+k
    l.append("s_mov_b32 exec_hi, 0")
    for k in range(L1):
        l.append(f"s_mov_b32 exec_lo, {hex(1 << (L0 + k))}")
        l.append(f"v_sub_f32 {W(L0 + k)}, {W(L0 + k)}, %[ts]")
    l.append("s_mov_b32 exec_lo, 0")
    l.append("s_mov_b32 exec_hi, 1")
    l.append(f"v_sub_f32 {W(L0 + L1)}, {W(L0 + L1)}, %[ts]")
    l.append("s_mov_b64 exec, %[sv]")
    return l


def shrink():
    """weights *= 1 - 3e-6 on every 1024th visit of a row (mixer.cpp:173-175); lanes whose row
    is not due multiply by exactly 1.0f."""
    return [f"v_pk_mul_f32 {W2(j)}, {W2(j)}, %[sc2]" for j in range(0, 4 * NQW, 2)]


def zero_rows():
    return [f"v_mov_b32 {W(j)}, 0" for j in range(4 * NQW)]


def main():
    blocks = {
        # part A: chunks every row has weights in (layer 1 / final: 33 weights = 9 chunks; what they
        # store beyond is padding that is zero in HBM and in the registers); part B: layer 0 only
        # layer-1 / final rows (144 bytes of weights, <= 9 lanes): lane-private pieces, as they always went
        "GMX_STK_LOAD_A": loads(0, NQA), "GMX_STK_STORE_A": stores(0, NQA), "GMX_STK_ADOPT_A": adopt(0, NQA),
        # ... and the rest of a layer-0 row the same way, for launches of few streams (and the per-bit
        # sessions): with the texture path to itself a wave moves its rows faster lane-private than through
        # the images (fewer instructions), with all four SIMDs of every CU doing it the images win
        "GMX_STK_LOAD_B": loads(NQA, NQW), "GMX_STK_STORE_B": stores(NQA, NQW), "GMX_STK_ADOPT_B": adopt(NQA, NQW),
        # launches of many streams: rows through staging images in LDS, coalesced on the HBM side -- with
        # fixed 33-row streams when most of them move, a few in a loop (fetch_sparse)
        "GMX_STK_FETCH_ROWS": fetch_rows(), "GMX_STK_FETCH_SPARSE": fetch_sparse(), "GMX_STK_EVICT_ROWS": evict_rows(),
        # lane <-> its image row: part A under the mask of every moving lane, part B under its layer-0 lanes
        "GMX_STK_TO_IMAGE_A": to_image(0, NQA), "GMX_STK_TO_IMAGE_B": to_image(NQA, NQW),
        "GMX_STK_FROM_IMAGE_A": from_image(0, NQA), "GMX_STK_FROM_IMAGE_B": from_image(NQA, NQW),
        "GMX_STK_FORWARD": forward(), "GMX_STK_FORWARD_EXACT": forward_exact(),
        "GMX_STK_LOAD_X": load_x_only(), "GMX_STK_OUTPUTS_TO_SGPRS": outputs_to_sgprs(),
        "GMX_STK_UPDATE": update(), "GMX_STK_SHRINK": shrink(), "GMX_STK_ZERO": zero_rows(),
    }
    out = "// Generated by gen_stock_asm.py -- do not edit.\n\n"
    for k, v in blocks.items():
        out += emit(k, v)
    out += f"#define GMX_STK_VGPR_LIMIT {RESERVED['v'][0]}\n"
    out += f"#define GMX_STK_ROW_PITCH {PITCH}\n"
    out += f"#define GMX_STK_SGPR_FIRST {RESERVED['s'][0]}\n"
    out += "// instructions: " + ", ".join(f"{k[8:].lower()} {len(v)}" for k, v in blocks.items()) + "\n"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gmx_stock_asm.inc")
    open(path, "w").write(out)
    print({k[8:].lower(): len(v) for k, v in blocks.items()})


if __name__ == "__main__":
    main()
