#!/usr/bin/env python3
"""Writes vexautonomousplanner_amd/csrc/vap_chain_asm.h: the chain wave's tile loops of the "wavefront of paths"
velocity kernel (k_velocity_lanes) as fully unrolled gfx950 assembly, one inline-asm statement per
(direction, tile length).

Why assembly, and why this shape (measured with tools/ubench_chain_lds.hip on MI355X):
  * a step of the recurrence is four dependent fp64 VALU instructions (32 cycles); its five coefficients come from LDS
    records {rho, g | am, A | cap} written by the producer waves;
  * an LDS instruction issued BETWEEN two dependent VALU instructions costs the wave ~15 cycles (104 cycles per step
    with three reads and a store in the shadow of the chain), issued back to back ~4 cycles: so the LDS traffic of NB
    steps is issued as one block (the previous batch's result stores, then the next batch's record reads into the other
    register bank), followed by the NB chain steps on registers, and one s_waitcnt per batch;
  * b128 reads need sub-register pairs, which inline-asm operands cannot name, so the banks are fixed registers
    (clobbers); the compiler keeps everything else out of them.

Record layout: the records of two consecutive slots (2m, 2m+1) of one path form an 80-byte PAIR (lane = path, lane
stride 80): +0 rho0, +8 g0, +16 am0, +24 A0, +32 rho1, +40 g1, +48 am1, +56 A1, +64 cap0, +72 cap1 — five b128 reads
per two steps (2.5 per step instead of 3 with one 48-byte record per slot: an LDS read costs the chain ~6.6 cycles).
Pair stride = P*80+64 bytes (STRIDE, an "i" operand bound to an assembler symbol so one text serves every P; the +64
keeps the producers' stores conflict-free).  Results: one double per step at oaddr + 8*slot.  Forward tiles consume
slots 0..TS-1, backward tiles TS-1..0.

    python3 tools/gen_chain_asm.py            # rewrites the header in place
"""
import os
import sys

NB = 4                      # steps per batch (register bank = NB records): banks of four fit v64-v151, so the kernel's
                            # waves fit 168 registers (three per SIMD); `--nb 8` writes the earlier banks of eight (v64-v239)
for _i, _a in enumerate(sys.argv):
    if _a == "--nb":
        NB = int(sys.argv[_i + 1])
FIRST = 36                  # first fixed register (banks of four: v36-v123, v127 with the sign-aware pairs: the kernel's waves fit 128)
BANK = [FIRST, FIRST + 10 * NB]   # first VGPR of bank A / B: NB*8 registers of {rho,g,am,A}, then NB*2 of cap
RES = FIRST + 20 * NB       # NB result pairs
LAST = RES + 2 * NB - 1
DUP_GN, DUP_C2 = LAST + 1, LAST + 3   # the sign-aware backward tiles: two more fixed register pairs
TILES = (16, 32, 64)


def rec(bank, k):
    b = BANK[bank] + 8 * k
    return dict(rho=f"v[{b}:{b+1}]", g=f"v[{b+2}:{b+3}]", am=f"v[{b+4}:{b+5}]", A=f"v[{b+6}:{b+7}]",
                lo=f"v[{b}:{b+3}]", hi=f"v[{b+4}:{b+7}]", cap=f"v[{BANK[bank]+8*NB+2*k}:{BANK[bank]+8*NB+2*k+1}]",
                cap2=f"v[{BANK[bank]+8*NB+2*k}:{BANK[bank]+8*NB+2*k+3}]")


def res(k):
    k %= NB
    return f"v[{RES+2*k}:{RES+2*k+1}]"


def loads(bank, first_slot):
    assert first_slot % 2 == 0 and NB % 2 == 0
    out = []
    for k in range(0, NB, 2):
        r0, r1 = rec(bank, k), rec(bank, k + 1)
        off = f"{(first_slot + k) // 2}*vap_stride_%="
        out.append(f"ds_read_b128 {r0['lo']}, %[addr] offset:{off}")
        out.append(f"ds_read_b128 {r0['hi']}, %[addr] offset:{off}+16")
        out.append(f"ds_read_b128 {r1['lo']}, %[addr] offset:{off}+32")
        out.append(f"ds_read_b128 {r1['hi']}, %[addr] offset:{off}+48")
        out.append(f"ds_read_b128 {r0['cap2']}, %[addr] offset:{off}+64")
    return out


def stores(first_slot):
    return [f"ds_write_b128 %[oaddr], v[{RES+4*m}:{RES+4*m+3}] offset:{8*(first_slot+2*m)}" for m in range(NB // 2)]


def chain(bank, backward, dup=False):
    out = []
    order = range(NB - 1, -1, -1) if backward else range(NB)
    if dup:
        # the sign-aware backward step (MPG:52-59; vap_device.h fast_backward_a<true>): a record with g < 0 marks a zero
        # heading difference, whose penalty is max(t, 0)*|g| instead of |t|*|g|.  With gn = (g < 0 ? 0 : g):
        #     clamp01(am - max(t, other)*|g|) = clamp01(min(fma(-t, |g|, am), fma(t, gn, am)))    bit for bit
        # (rounding is monotone; k_velocity_chase's bwd_step_dup2).  The gn of a bank's records do not depend on the
        # state: they are formed in a block ahead of the dependent steps (three instructions per record, in place in
        # two fixed registers past the banks, v[DUP_GN:DUP_GN+1]; the second FMA's result in v[DUP_C2:DUP_C2+1]).
        assert backward
    for k in order:
        r = rec(bank, k)
        u, up = (res(k + 1), res(k + 2)) if backward else (res(k - 1), res(k - 2))
        if dup:
            g_lo, g_hi = r['g'][2:-1].split(':')
            out.append(f"v_cmp_gt_f64 vcc, 0, {r['g']}")
            out.append(f"v_cndmask_b32_e64 v{DUP_GN}, v{g_lo}, 0, vcc")
            out.append(f"v_cndmask_b32_e64 v{DUP_GN+1}, v{g_hi}, 0, vcc")
            out.append(f"v_fma_f64 %[t], -{r['rho']}, {up}, {u}")
            out.append(f"v_fma_f64 v[{DUP_C2}:{DUP_C2+1}], %[t], v[{DUP_GN}:{DUP_GN+1}], {r['am']}")
            out.append(f"v_fma_f64 %[t], -%[t], |{r['g']}|, {r['am']}")
            out.append(f"v_min_f64 %[t], %[t], v[{DUP_C2}:{DUP_C2+1}] clamp")
            out.append(f"v_fma_f64 %[t], {r['A']}, %[t], {u}")
            out.append(f"v_min_f64 {res(k)}, %[t], {r['cap']}")
            continue
        out.append(f"v_fma_f64 %[t], -{r['rho']}, {up}, {u}")
        out.append(f"v_fma_f64 %[t], -|%[t]|, |{r['g']}|, {r['am']} clamp")
        out.append(f"v_fma_f64 %[t], {r['A']}, %[t], {u}")
        out.append(f"v_min_f64 {res(k)}, %[t], {r['cap']}")
    return out


def tile(ts, backward, split, dup=False):
    nb = ts // NB
    slots = [(nb - 1 - n) * NB if backward else n * NB for n in range(nb)]   # first slot of batch n
    u_reg, up_reg = (RES, RES + 2) if backward else (RES + 2 * (NB - 1), RES + 2 * (NB - 2))
    a = [".set vap_stride_%=, %[stride]",
         f"v_mov_b32 v{u_reg}, %[ulo]", f"v_mov_b32 v{u_reg+1}, %[uhi]",
         f"v_mov_b32 v{up_reg}, %[plo]", f"v_mov_b32 v{up_reg+1}, %[phi]"]
    a += loads(0, slots[0])
    a.append("s_waitcnt lgkmcnt(0)")
    for n in range(nb):
        bank = n & 1
        lds = (stores(slots[n - 1]) if n > 0 else []) + (loads(bank ^ 1, slots[n + 1]) if n + 1 < nb else [])
        ch = chain(bank, backward, dup)
        if split and len(lds) > 15:
            # two LDS blocks of at most 15 operations (the lgkmcnt counter's range), each in front of half the steps
            h = 15 if len(lds) - 15 <= 15 else len(lds) // 2
            a += lds[:h] + ch[:len(ch) // 2] + lds[h:] + ch[len(ch) // 2:]
        else:
            a += lds + ch
        a.append("s_waitcnt lgkmcnt(0)")
    a += stores(slots[nb - 1])
    a += [f"v_mov_b32 %[ulo], v{u_reg}", f"v_mov_b32 %[uhi], v{u_reg+1}",
          f"v_mov_b32 %[plo], v{up_reg}", f"v_mov_b32 %[phi], v{up_reg+1}"]
    return a


def function(name, ts, backward, split, dup=False):
    lines = tile(ts, backward, split, dup)
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{i}"' for i in range(BANK[0], (DUP_C2 + 2) if dup else (LAST + 1)))
    if dup:
        clob = '"vcc", ' + clob
    # wrap the clobber list
    words = clob.split(", ")
    clob = ",\n          ".join(", ".join(words[i:i + 12]) for i in range(0, len(words), 12))
    return f"""template <int STRIDE>
__device__ __forceinline__ void {name}(uint32_t addr, uint32_t oaddr, double &u, double &up)
{{
    uint32_t ulo = (uint32_t)__builtin_bit_cast(uint64_t, u), uhi = (uint32_t)(__builtin_bit_cast(uint64_t, u) >> 32);
    uint32_t plo = (uint32_t)__builtin_bit_cast(uint64_t, up), phi = (uint32_t)(__builtin_bit_cast(uint64_t, up) >> 32);
    double t;
    asm volatile(
{body}
        : [ulo] "+v"(ulo), [uhi] "+v"(uhi), [plo] "+v"(plo), [phi] "+v"(phi), [t] "=&v"(t)
        : [addr] "v"(addr), [oaddr] "v"(oaddr), [stride] "i"(STRIDE)
        : "memory",
          {clob});
    u = __builtin_bit_cast(double, ((uint64_t)uhi << 32) | ulo);
    up = __builtin_bit_cast(double, ((uint64_t)phi << 32) | plo);
}}
"""


# ------------------------------------------------------------------------------------------------------------------
# The ROTATING chain (16 paths per workgroup, tile = 64 samples): a path owns a QUAD of lanes (lane = 4*path + r), and
# the recurrence's state travels round the quad — lane r walks the four consecutive steps 16m+4r .. 16m+4r+3 of group
# m, then hands (u, u_prev) to lane r+1 by DPP (v_mov_b32 quad_perm, four moves).  Why: an LDS instruction costs the
# chain wave the same whether 16 or 64 of its lanes carry a path; with every lane reading the records of ITS four steps,
# one block of ten 16-byte reads + two result stores serves 16 steps instead of 4 (0.75 LDS instructions per step
# instead of 4): the ~24 cycles per step that LDS traffic adds to the 32 of the four dependent instructions shrink to
# ~4, for ~3 of DPP hand-over.  A turn runs under EXEC = the quad lanes whose turn it is (the other lanes' registers —
# their own results — stay as they are), the hand-over under full EXEC.
#   records: group m (16 steps), plane k (the lane's first / second pair), lane L: m*GROUP + k*PLANE + L*80, the 80-byte
#   pair record as above; results: lane L's four doubles at oaddr + 8*(16m + 4r) (oaddr = the path's row + 32*r).
ROT0 = 24                                   # first fixed register
ROT_BANK = [ROT0, ROT0 + 40]                # two banks of two pair records (20 registers each)
ROT_R = ROT0 + 80                           # R0..R3: the lane's four results (8 registers)
ROT_U, ROT_UP = ROT0 + 88, ROT0 + 90
ROT_GN, ROT_C2 = ROT0 + 92, ROT0 + 94       # sign-aware tiles
ROT_LAST = ROT0 + 95


def rot_rec(bank, i):
    """registers of step i (0..3) of the lane's turn in bank `bank`"""
    k, j = i >> 1, i & 1
    b = ROT_BANK[bank] + 20 * k
    return dict(rho=f"v[{b+8*j}:{b+8*j+1}]", g=f"v[{b+8*j+2}:{b+8*j+3}]", am=f"v[{b+8*j+4}:{b+8*j+5}]",
                A=f"v[{b+8*j+6}:{b+8*j+7}]", cap=f"v[{b+16+2*j}:{b+16+2*j+1}]", g_lo=b + 8 * j + 2, g_hi=b + 8 * j + 3)


def rot_loads(bank, m):
    out = []
    for k in range(2):
        b = ROT_BANK[bank] + 20 * k
        off = f"{m}*vap_group_%=+{k}*vap_plane_%="
        for w in range(5):
            out.append(f"ds_read_b128 v[{b+4*w}:{b+4*w+3}], %[addr] offset:{off}+{16*w}")
    return out


def rot_stores(m):
    return [f"ds_write_b128 %[oaddr], v[{ROT_R}:{ROT_R+3}] offset:{128*m}",
            f"ds_write_b128 %[oaddr], v[{ROT_R+4}:{ROT_R+7}] offset:{128*m+16}"]


def rot_turn(bank, q, backward, dup):
    R = lambda i: f"v[{ROT_R+2*i}:{ROT_R+2*i+1}]"
    U, UP = f"v[{ROT_U}:{ROT_U+1}]", f"v[{ROT_UP}:{ROT_UP+1}]"
    out = [f"s_mov_b64 exec, %[m{q}]"]
    order = (3, 2, 1, 0) if backward else (0, 1, 2, 3)
    for n, i in enumerate(order):
        r = rot_rec(bank, i)
        prev = order[n - 1] if n >= 1 else None
        prev2 = order[n - 2] if n >= 2 else None
        u_in = U if n == 0 else R(prev)
        up_in = UP if n == 0 else (U if n == 1 else R(prev2))
        if dup:
            out.append(f"v_cmp_gt_f64 vcc, 0, {r['g']}")
            out.append(f"v_cndmask_b32_e64 v{ROT_GN}, v{r['g_lo']}, 0, vcc")
            out.append(f"v_cndmask_b32_e64 v{ROT_GN+1}, v{r['g_hi']}, 0, vcc")
            out.append(f"v_fma_f64 %[t], -{r['rho']}, {up_in}, {u_in}")
            out.append(f"v_fma_f64 v[{ROT_C2}:{ROT_C2+1}], %[t], v[{ROT_GN}:{ROT_GN+1}], {r['am']}")
            out.append(f"v_fma_f64 %[t], -%[t], |{r['g']}|, {r['am']}")
            out.append(f"v_min_f64 %[t], %[t], v[{ROT_C2}:{ROT_C2+1}] clamp")
            out.append(f"v_fma_f64 %[t], {r['A']}, %[t], {u_in}")
            out.append(f"v_min_f64 {R(i)}, %[t], {r['cap']}")
        else:
            out.append(f"v_fma_f64 %[t], -{r['rho']}, {up_in}, {u_in}")
            out.append(f"v_fma_f64 %[t], -|%[t]|, |{r['g']}|, {r['am']} clamp")
            out.append(f"v_fma_f64 %[t], {r['A']}, %[t], {u_in}")
            out.append(f"v_min_f64 {R(i)}, %[t], {r['cap']}")
    last, last2 = order[3], order[2]
    perm = "quad_perm:[1,2,3,0]" if backward else "quad_perm:[3,0,1,2]"     # lane i takes from lane i+1 / i-1 of its quad
    out.append("s_mov_b64 exec, -1")
    # (the previous-but-one result first: the DPP read of the last result then has its two wait states)
    out.append(f"v_mov_b32_dpp v{ROT_UP}, v{ROT_R+2*last2} {perm} row_mask:0xf bank_mask:0xf")
    out.append(f"v_mov_b32_dpp v{ROT_UP+1}, v{ROT_R+2*last2+1} {perm} row_mask:0xf bank_mask:0xf")
    out.append(f"v_mov_b32_dpp v{ROT_U}, v{ROT_R+2*last} {perm} row_mask:0xf bank_mask:0xf")
    out.append(f"v_mov_b32_dpp v{ROT_U+1}, v{ROT_R+2*last+1} {perm} row_mask:0xf bank_mask:0xf")
    return out


def rot_tile(backward, dup):
    groups = [3, 2, 1, 0] if backward else [0, 1, 2, 3]
    turns = [3, 2, 1, 0] if backward else [0, 1, 2, 3]
    a = [".set vap_group_%=, %[group]", ".set vap_plane_%=, %[plane]",
         f"v_mov_b32 v{ROT_U}, %[ulo]", f"v_mov_b32 v{ROT_U+1}, %[uhi]",
         f"v_mov_b32 v{ROT_UP}, %[plo]", f"v_mov_b32 v{ROT_UP+1}, %[phi]"]
    a += rot_loads(0, groups[0])
    a.append("s_waitcnt lgkmcnt(0)")
    for n, m in enumerate(groups):
        bank = n & 1
        # one LDS block per group: the previous group's results out, the next group's records in
        st = rot_stores(groups[n - 1]) if n > 0 else []
        ld = rot_loads(bank ^ 1, groups[n + 1]) if n + 1 < 4 else []
        # ... in two parts (the lgkmcnt counter counts to 15), each issued under full EXEC: the stores (the first turns
        # overwrite their lanes' results) and half the reads before the first turn, the other reads between the second
        # turn's hand-over and the third turn
        a += st + ld[:len(ld) // 2]
        a += rot_turn(bank, turns[0], backward, dup) + rot_turn(bank, turns[1], backward, dup)
        a += ld[len(ld) // 2:]
        a += rot_turn(bank, turns[2], backward, dup) + rot_turn(bank, turns[3], backward, dup)
        a.append("s_waitcnt lgkmcnt(0)")
    a += rot_stores(groups[3])
    a += [f"v_mov_b32 %[ulo], v{ROT_U}", f"v_mov_b32 %[uhi], v{ROT_U+1}",
          f"v_mov_b32 %[plo], v{ROT_UP}", f"v_mov_b32 %[phi], v{ROT_UP+1}"]
    return a


def rot_function(name, backward, dup=False):
    lines = rot_tile(backward, dup)
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    words = [f'"v{i}"' for i in range(ROT0, (ROT_LAST + 1) if dup else ROT_GN)]
    if dup:
        words = ['"vcc"'] + words
    clob = ",\n          ".join(", ".join(words[i:i + 12]) for i in range(0, len(words), 12))
    return f"""// (the wave enters and leaves with all 64 lanes enabled; m0..m3 = the lanes with lane % 4 == 0..3)
template <int GROUP, int PLANE>
__device__ __forceinline__ void {name}(uint32_t addr, uint32_t oaddr, double &u, double &up, uint64_t m0, uint64_t m1, uint64_t m2,
                                       uint64_t m3)
{{
    uint32_t ulo = (uint32_t)__builtin_bit_cast(uint64_t, u), uhi = (uint32_t)(__builtin_bit_cast(uint64_t, u) >> 32);
    uint32_t plo = (uint32_t)__builtin_bit_cast(uint64_t, up), phi = (uint32_t)(__builtin_bit_cast(uint64_t, up) >> 32);
    double t;
    asm volatile(
{body}
        : [ulo] "+v"(ulo), [uhi] "+v"(uhi), [plo] "+v"(plo), [phi] "+v"(phi), [t] "=&v"(t)
        : [addr] "v"(addr), [oaddr] "v"(oaddr), [group] "i"(GROUP), [plane] "i"(PLANE), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2),
          [m3] "s"(m3)
        : "memory",
          {clob});
    u = __builtin_bit_cast(double, ((uint64_t)uhi << 32) | ulo);
    up = __builtin_bit_cast(double, ((uint64_t)phi << 32) | plo);
}}
"""


ROT_GEOMETRY = f"""
// ---- the rotating chain (16 paths per workgroup, tile = 64 samples): a path owns a quad of lanes and the state travels
// round it (tools/gen_chain_asm.py).  Records: group m (16 steps), plane k (the lane's first / second pair of steps),
// chain lane L = 4*path + r (r = which four steps of the group): m*kRotGroup + k*kRotPlane + L*80.
constexpr int kRotPlane = 64 * 80 + 64;        // (+64: the producers' stores of a pair and of the next land in different banks)
constexpr int kRotGroup = 2 * kRotPlane;
constexpr int kRotTileBytes = 4 * kRotGroup;
constexpr int kRotFirstVgpr = {ROT0}, kRotLastVgpr = {ROT_LAST};
__host__ __device__ constexpr int rot_rec_off(int p, int s)   // byte offset of {{rho, g | am, A}} of sample s of path p in a tile
{{
    return ((s >> 4) * kRotGroup) + (((s >> 1) & 1) * kRotPlane) + ((4 * p + ((s >> 2) & 3)) * 80) + ((s & 1) * 32);
}}
__host__ __device__ constexpr int rot_cap_off(int p, int s)
{{
    return ((s >> 4) * kRotGroup) + (((s >> 1) & 1) * kRotPlane) + ((4 * p + ((s >> 2) & 3)) * 80) + 64 + ((s & 1) * 8);
}}

"""

HEADER = f"""// vap_chain_asm.h — GENERATED by tools/gen_chain_asm.py (edit the generator, not this file).
//
// The chain wave's tile loops of k_velocity_lanes ("a wavefront of paths": lane = path), fully unrolled gfx950
// assembly.  chain_fwd_<TS> walks slots 0..TS-1 of a tile of LDS records, chain_bwd_<TS> walks TS-1..0; the records of
// slots 2m, 2m+1 are one 80-byte pair {{rho0, g0 | am0, A0 | rho1, g1 | am1, A1 | cap0, cap1}} (lane stride 80, pair stride
// STRIDE bytes); one double per slot goes to oaddr + 8*slot.  u / up are the last two squared velocities (MPG:188-311
// in the scaled four-instruction form of vap_device.h step4, bit for bit).  Registers v{BANK[0]}..v{LAST} are the two
// record banks and the result pairs; chain_bwd_dup_<TS> (the sign-aware step, MPG:52-59) also uses v{DUP_GN}..v{DUP_C2+1}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vap {{

constexpr int kChainBatch = {NB};          // steps per register bank
constexpr int kChainFirstVgpr = {BANK[0]}, kChainLastVgpr = {LAST};

"""


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    split = "--no-split" not in sys.argv
    out = HEADER
    for ts in TILES:
        out += function(f"chain_fwd_{ts}", ts, False, split)
        out += "\n"
        out += function(f"chain_bwd_{ts}", ts, True, split)
        out += "\n"
        out += function(f"chain_bwd_dup_{ts}", ts, True, split, True)
        out += "\n"
    out += ROT_GEOMETRY
    out += rot_function("chain_rot_fwd", False)
    out += "\n"
    out += rot_function("chain_rot_bwd", True)
    out += "\n"
    out += rot_function("chain_rot_bwd_dup", True, True)
    out += "\n"
    out += """// dispatch on the tile length
template <int STRIDE, int TS>
__device__ __forceinline__ void chain_fwd(uint32_t addr, uint32_t oaddr, double &u, double &up)
{
    static_assert(TS == 16 || TS == 32 || TS == 64, "tile lengths the generator wrote");
    if constexpr (TS == 16) chain_fwd_16<STRIDE>(addr, oaddr, u, up);
    else if constexpr (TS == 32) chain_fwd_32<STRIDE>(addr, oaddr, u, up);
    else chain_fwd_64<STRIDE>(addr, oaddr, u, up);
}
template <int STRIDE, int TS>
__device__ __forceinline__ void chain_bwd(uint32_t addr, uint32_t oaddr, double &u, double &up)
{
    static_assert(TS == 16 || TS == 32 || TS == 64, "tile lengths the generator wrote");
    if constexpr (TS == 16) chain_bwd_16<STRIDE>(addr, oaddr, u, up);
    else if constexpr (TS == 32) chain_bwd_32<STRIDE>(addr, oaddr, u, up);
    else chain_bwd_64<STRIDE>(addr, oaddr, u, up);
}
// ... a tile that holds a sample with a zero heading difference (g < 0): the sign-aware step on every slot
template <int STRIDE, int TS>
__device__ __forceinline__ void chain_bwd_dup(uint32_t addr, uint32_t oaddr, double &u, double &up)
{
    static_assert(TS == 16 || TS == 32 || TS == 64, "tile lengths the generator wrote");
    if constexpr (TS == 16) chain_bwd_dup_16<STRIDE>(addr, oaddr, u, up);
    else if constexpr (TS == 32) chain_bwd_dup_32<STRIDE>(addr, oaddr, u, up);
    else chain_bwd_dup_64<STRIDE>(addr, oaddr, u, up);
}

}  // namespace vap
"""
    path = sys.argv[-1] if sys.argv[-1].endswith(".h") else os.path.join(root, "vexautonomousplanner_amd", "csrc", "vap_chain_asm.h")
    with open(path, "w") as f:
        f.write(out)
    print(f"wrote {path}: NB={NB}, registers v{BANK[0]}..v{LAST}, split={split}")


if __name__ == "__main__":
    main()
