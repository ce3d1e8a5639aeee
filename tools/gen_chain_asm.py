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
BANK = [64, 64 + 10 * NB]   # first VGPR of bank A / B: NB*8 registers of {rho,g,am,A}, then NB*2 of cap
RES = 64 + 20 * NB          # NB result pairs
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
