// di_unit_kernel_r5.hip -- NOT BUILT, NOT SHIPPED: the unit-speed segment kernel of round 5's config-4 trial, kept as text
// for whoever takes the store problem up again.  It lived in csrc/dense_batched.hip after di_segment_kernel (whose DiParams, DiState,
// slot_totals16, PDMP_ROW_STORE it uses) and was launched before it with a per-chain `general` flag array appended to the workspace;
// di_segment_kernel skipped the chains the flag left to this kernel.  Measured on one MI355X, d = 1000, 10^4 chains x 200:
// 4.74 ms without row stores and column loads (di_segment_kernel: 6.09), 7.93 ms with them (7.40).
// ---------------------------------------------------------------- the same segment for unit speeds, |v_i| = 1 (round 5)
// Every velocity the reference's ZigZag constructors and init_state produce is +-1 (AbstractPDMP.jl:93-153), and a flip keeps |v_i|.
// Then v is one BIT per coordinate, and what the bound, the proposal and the Categorical walk read is never P x or P v but their signed
// products  a_i = (P x)_i v_i,  b_i = (P v)_i v_i  (multiplying by +-1 is exact, so carrying a, b instead of P x, P v changes no
// rounding):
//   rate of coordinate i along the flow       a_i + t b_i               one fma (was fma + mul)
//   flip of coordinate m                       b_i -= (2 v_m v_i) P_im   one fma with a +-2 built from two sign bits; then a_m, b_m change sign
// and the right end of a cell of the bound is the left end of the next one (the same grid time, the same fma), kept in a register.
// Flows cost nothing: a is kept at the local time `off` of the last PROPOSAL (rate_i(t) = a_i + (t - off) b_i); a proposal at tp moves a
// there (the fma it needs anyway), an accepted one makes tp the new origin (off = 0), a move to the horizon shifts the origin (off -= tau).
// x is needed by nothing but the records: it stays in LDS and is flowed ONCE per record, by the time elapsed since the last one
// (x + v (tau_1 + tau_2) instead of (x + v tau_1) + v tau_2: one rounding less, inside this path's tolerance).
// Every wave-wide total is passed through v_readfirstlane (uni): the compiler cannot know that a DPP / permlane reduction ends with the
// same value in every lane, and without that every branch of the state machine is compiled as a divergent one (exec masks, counters
// and times in vector registers, v_cndmask merges).
// Registers: a, b (64) + at most one of {cell ends, column of P, slot rates} (32) alive at a time: 3 waves per SIMD instead of 2, and
// 4 instead of 7 instructions per coordinate and cell.  A chain with any |v_i| != 1 is left to di_segment_kernel (flag per chain).
#ifndef UNIT_WAVES
#define UNIT_WAVES 2
#endif
__device__ __forceinline__ double with_sign_bit(double mag_nonneg, uint32_t bits, int s) {        // sign <- bit s of `bits`
    const uint32_t hi = (uint32_t)__double2hiint(mag_nonneg);
    const uint32_t sh = bits << (31 - s);
    return __hiloint2double((int)((sh & 0x80000000u) | (hi & 0x7fffffffu)), __double2loint(mag_nonneg));     // v_bfi_b32
}
__device__ __forceinline__ double flip_by_bit(double val, uint32_t bits, int s) {                  // val * (bit s of `bits` ? -1 : 1)
    const uint32_t sh = bits << (31 - s);
    return __hiloint2double((int)((uint32_t)__double2hiint(val) ^ (sh & 0x80000000u)), __double2loint(val));
}
__device__ __forceinline__ double vmax(double x, double y) {   // v_max_f64 as such: fmax() of a value the compiler has not seen produced (a loop-carried
    double r;                                                  // register, say) comes with a canonicalising v_max_f64 x, x in front
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ double vmax0(double x) { double r; asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ double uni(double x) {           // a wave-uniform double, known to be one (two v_readfirstlane)
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
// A row of d <= 1024 doubles read or written by a wave, slot s = coordinates s * 64 + lane: one scalar base per eight slots (the immediate
// offset of a global access reaches 4 KB), hidden from the compiler -- which would otherwise keep one vector offset per slot >= 8 alive
// for the whole kernel -- and typed as a global pointer (a pointer that has been through an asm is a flat one to the compiler: flat
// loads, 64-bit vector addresses, both counters).
template <class T> struct Row {
    using G = __attribute__((address_space(1))) T;
    G *lo, *hi;
    __device__ __forceinline__ explicit Row(T *q) {
        lo = (G *)q;
        uint64_t h = (uint64_t)(q + 8 * WAVE);
        asm("" : "+s"(h));
        hi = (G *)h;
    }
    __device__ __forceinline__ G *ptr(int s, int lane) const { return (s < 8 ? lo : hi) + ((s & 7) * WAVE + lane); }
};
// Which lanes of slot s hold a coordinate (i = 64 s + lane < d): all of them in the lower half of the slots (a kernel with KC >= 8 slots
// is launched for d > 32 KC only), and above that all, some (the one ragged slot) or none -- a SCALAR question first, so that the full
// slots run without an exec mask (sixteen lane masks in scalar registers were the other half of the register spills).
template <int KC, class F> __device__ __forceinline__ void on_slot(int s, int d, int d_opaque, int lane, F &&f) {
    if (KC >= 8 && s < KC / 2) { f(); return; }
    asm volatile("" : "+s"(d_opaque));        // (asked again at every use: hoisted out of the event loop, the answers are eight lane masks in scalar registers)
    if (d_opaque - s * WAVE >= WAVE) f();
    else if (lane < d - s * WAVE) f();
}
__device__ __forceinline__ uint64_t uni_bits(double x) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint(x)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(__double2loint(x));
}
template <int KC, bool UNTIL = false>
__global__ __launch_bounds__(256, UNIT_WAVES) void di_unit_kernel(const DiParams p, int32_t *__restrict__ general) {
    const int lane = lane_id();
    const int64_t j = p.c_lo + (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (j >= p.c_hi) return;
    const int64_t c = (UNTIL && p.chain_map) ? p.chain_map[j] : j;
    const bool first = p.k_begin == 0 && !(UNTIL && p.resume);
    if (!first && general[c]) return;                  // (set by this kernel in the chain's first segment)
    DiState cs;
    if (first) { cs.t = 0.0; cs.horizon = p.tmax; cs.rng_n = 0; cs.dead = 0; cs.pad = 0; }
    else cs = p.st[c];
    if (cs.dead || (UNTIL && cs.pad && p.k_begin != 0)) return;
    const int d = p.d;
    int dq = d;
    asm("" : "+s"(dq));                                // (d again, as a value the compiler does not connect with d: on_slot)
    const bool adaptive = p.adaptive != 0;
    const int dbg = p.pad;                                // PDMP_DENSE_DEBUG (measurements only): 1 = no row stores, 2 = no loads of P's columns, 4 = plain instead of streaming stores, 8 = no stores of V
    __shared__ double x_lds[4][KC * WAVE];
    double *x = x_lds[threadIdx.x >> 6] + lane;          // slot s at x[s * WAVE]: the position at the chain's last record
    double a[KC], b[KC];
    uint32_t sb = 0u;                                     // bit s: v of coordinate s * WAVE + lane is -1
    {
        const Row<const double> xin(p.In + (size_t)c * d), vin(p.In + ((size_t)p.C + (size_t)c) * d);
        const Row<const double> gxin(p.Out + (size_t)c * d), gvin(p.Out + ((size_t)p.C + (size_t)c) * d);
        bool unit = true;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            double vi = 1.0;
            on_slot<KC>(s, d, dq, lane, [&] { vi = *vin.ptr(s, lane); });
            unit = unit && (vi == 1.0 || vi == -1.0);
            if (vi < 0.0) sb |= 1u << s;
        }
        if (first) {
            const bool all_unit = __ballot(!unit) == 0ull;
            if (lane == 0) general[c] = all_unit ? 0 : 1;
            if (!all_unit) return;
        }
#pragma unroll
        for (int s = 0; s < KC; ++s) { double xi = 0.0; on_slot<KC>(s, d, dq, lane, [&] { xi = *xin.ptr(s, lane); }); x[s * WAVE] = xi; }
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            a[s] = 0.0; b[s] = 0.0;
            on_slot<KC>(s, d, dq, lane, [&] { a[s] = flip_by_bit(*gxin.ptr(s, lane), sb, s); b[s] = flip_by_bit(*gvin.ptr(s, lane), sb, s); });
        }
    }
    ChainRng rng;
    rng.init(p.seed, (uint64_t)(p.chain0 + c));
    rng.restore(cs.rng_n);
    double t = cs.t, horizon = cs.horizon, ts = 0.0, ar = 0.0, E = 0.0, hB = horizon, el = 0.0, off = 0.0;
    uint64_t ev[5] = {0, 0, 0, 0, 0};                   // error_value_ar, as bits (five scalar register pairs; as doubles they sit in vector registers)
    int errored = 0, rejected = 0, hits = 0, mode = DB_FRESH;
    int64_t k = p.k_begin, it = 0;
    bool dead = false;

    // A record in two steps, because of the one load the loop makes (a column of P at a flip): on gfx9 loads and stores share one
    // counter (vmcnt), and waiting for a load while stores are in flight means waiting for all of them.  With the stores of a record
    // issued BEFORE the column is consumed, every event drained its 16 KB of row stores before it went on (the round-4 kernel did);
    // now: loads issued, x flowed in LDS (flow_x), column consumed, THEN the stores (store_rows) -- they drain during the next event.
    auto flow_x = [&](uint32_t sbx) {                     // x <- x + v el, along the velocity x had on the way (sbx)
#pragma unroll
        for (int s = 0; s < KC; ++s) x[s * WAVE] = x[s * WAVE] + flip_by_bit(el, sbx, s);
        el = 0.0;
    };
    auto store_rows = [&](int64_t kk) {
        const size_t e = (size_t)j * (size_t)p.n_sk + (size_t)kk;
        const Row<double> Xe(p.X + e * (size_t)d), Ve(p.V + e * (size_t)d);
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            if (dbg & 1) break;
            on_slot<KC>(s, d, dq, lane, [&] {
                if (dbg & 4) { *Xe.ptr(s, lane) = x[s * WAVE]; if (!(dbg & 8)) *Ve.ptr(s, lane) = with_sign_bit(1.0, sb, s); }
                else { PDMP_ROW_STORE(Xe.ptr(s, lane), x[s * WAVE]); if (!(dbg & 8)) PDMP_ROW_STORE(Ve.ptr(s, lane), with_sign_bit(1.0, sb, s)); }
            });
        }
        if (lane == 0) {
            p.t[e] = t;
            if (p.H) p.H[e] = horizon;
            if (p.AR) p.AR[e] = ar;
            if (p.EB) p.EB[e] = errored;
            if (p.RJ) p.RJ[e] = rejected;
            if (p.HH) p.HH[e] = hits;
            if (p.EV) for (int q = 0; q < 5; ++q) p.EV[5 * e + q] = __longlong_as_double((long long)ev[q]);
        }
    };
    auto flow = [&](double tau) { el = uni(el + tau); off = uni(off - tau); };     // the origin moves; a stays where it is

    if (k == 0) { flow_x(sb); store_rows(0); k = 1; }
    while (k < p.k_end && (!UNTIL || t < p.t_stop)) {
        if (++it > p.cap * 4) { dead = true; break; }
        const double harg = uni((mode == DB_ERR) ? horizon / 2 : ((mode == DB_REUSE) ? hB : horizon));
        hB = harg;
        const double h = uni(harg / 9.0);
        const double e1 = uni(rng.exponential());        // (the logarithm is a call: its result is a divergent value to the compiler)
        E = uni((mode == DB_REUSE) ? E + e1 : e1);
        double acc = 0.0, cprev = 0.0, tp = __builtin_huge_val(), lb = 0.0;
        // the rate at the right end of a cell is the rate at the left end of the next: kept in registers.  The nine cells are unrolled
        // (as a loop the carried ends cost sixteen register copies per cell, or a second set of thirty-two registers)
        double end[KC];
        { const double d0 = 0.0 - off;
#pragma unroll
          for (int s = 0; s < KC; ++s) end[s] = fma(d0, b[s], a[s]); }
        auto cell = [&](const int jc) __attribute__((always_inline)) -> bool {          // true: the event falls into this cell
            const double t0 = (double)jc * h, t1 = (jc == 8) ? harg : (double)(jc + 1) * h;
            const double d1 = t1 - off;
            double part;
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                const double r = fma(d1, b[s], a[s]);
                const double cm = vmax0(vmax(end[s], r));                 // "affine cell": max(v0, v1, 0)
                end[s] = r;
                part = (s == 0) ? cm : part + cm;
            }
            const double bj = uni(wave_sum(part));
            acc = uni(acc + bj);
            const double cj = uni(acc * h);
            lb = bj;
            if (!(cj < E)) { tp = uni(t0 + (E - cprev) / (cj - cprev) * h); return true; }
            cprev = cj;
            return false;
        };
        (void)(cell(0) || cell(1) || cell(2) || cell(3) || cell(4) || cell(5) || cell(6) || cell(7) || cell(8));
        bool propose;
        if (mode == DB_ERR) {
            if (adaptive) horizon = hB;
            errored++;
            { const int slot = errored % 5;
#pragma unroll
              for (int q = 0; q < 5; ++q) if (q == slot) ev[q] = uni_bits(ar); }
            propose = tp < horizon;
        } else if (mode == DB_REUSE) {
            if (adaptive) horizon = uni(horizon / 1.04);
            rejected++;
            if (tp > horizon) { flow(horizon); ts = uni(ts + horizon); hits++; }
            propose = tp < horizon;
        } else if (tp > horizon) {
            flow(horizon);
            ts = uni(ts + horizon);
            hits++;
            if (adaptive) horizon = uni(horizon * 1.01);
            propose = false;
        } else {
            propose = tp < horizon;
        }
        mode = DB_FRESH;
        if (!propose) continue;
        double part = 0.0;
        { const double dp = tp - off;                     // a moves to the proposal's time
#pragma unroll
          for (int s = 0; s < KC; ++s) { a[s] = fma(dp, b[s], a[s]); const double r = fmax(0.0, a[s]); part = (s == 0) ? r : part + r; } }
        off = tp;
        const double lam = uni(wave_sum(part));
        ar = uni(lam / lb);
        if (ar > 1.0) { mode = DB_ERR; continue; }
        const double u = uni(rng.uniform());
        if (!(u < ar)) { mode = DB_REUSE; continue; }
        flow(tp);                                         // accepted: tp is the origin (off = 0 exactly)
        const double u2 = uni(rng.uniform());
        if (lam > 0.0) {
            const double theta = uni(u2 * lam);
            int sh = -1;
            double carry = 0.0, carry_at = 0.0;
            if constexpr (KC == 16) {
                double lv[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) lv[s] = fmax(0.0, a[s]);
                const double tot = slot_totals16(lv, lane);
                const double Ts = __shfl(tot, slot_lane16(lane & 15), WAVE);
                double pre = Ts;
                { double q;
                  q = dpp_d<DPP_ROW_SHR1>(pre); if ((lane & 15) >= 1) pre = pre + q;
                  q = dpp_d<DPP_ROW_SHR2>(pre); if ((lane & 15) >= 2) pre = pre + q;
                  q = dpp_d<DPP_ROW_SHR4>(pre); if ((lane & 15) >= 4) pre = pre + q;
                  q = dpp_d<DPP_ROW_SHR8>(pre); if ((lane & 15) >= 8) pre = pre + q; }
                const unsigned hit16 = (unsigned)(__ballot(pre > theta) & 0xffffull);
                if (hit16 != 0u) {
                    sh = (int)__builtin_ctz(hit16);
                    carry_at = (sh == 0) ? 0.0 : readlane_d(pre, sh - 1);
                }
            } else {
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                if (sh < 0) {
                    const double Ts = uni(wave_sum(fmax(0.0, a[s])));
                    if (carry + Ts > theta) { sh = s; carry_at = carry; }
                    carry = carry + Ts;
                }
            }
            }
            int m = -1;
            if (sh >= 0) {
                double lsel = 0.0;
#pragma unroll
                for (int s = 0; s < KC; ++s) if (s == sh) lsel = fmax(0.0, a[s]);
                const double inc = wave_scan_incl(lsel, lane);
                const unsigned long long hit = __ballot((sh * WAVE + lane < d) && (carry_at + inc > theta));
                if (hit != 0ull) m = sh * WAVE + (int)__builtin_ctzll(hit);
            }
            if (m < 0) m = d - 1;
            m = __builtin_amdgcn_readfirstlane(m);
            const int ms = m >> 6, ml = m & (WAVE - 1);
            const uint32_t sbm = (uint32_t)__builtin_amdgcn_readlane((int)sb, ml);
            const uint32_t sx = sb ^ (((sbm >> ms) & 1u) ? 0xffffffffu : 0u);        // bit s: v_m v_i = -1 (before the flip)
            const Row<const double> Pm(p.P + (size_t)m * (size_t)d);
            double pc[KC];                      // column m of P: issued here, consumed after x has been flowed in LDS
#pragma unroll
            for (int s = 0; s < KC; ++s) { pc[s] = 0.0; if (!(dbg & 2)) on_slot<KC>(s, d, dq, lane, [&] { pc[s] = *Pm.ptr(s, lane); }); }
            const uint32_t sb_before = sb;
            if (lane == ml) sb ^= 1u << ms;                                          // v[m] *= -1
#pragma unroll
            for (int s = 0; s < KC; ++s) if (s == ms && lane == ml) a[s] = -a[s];
            t = uni(t + tp + ts);
            ts = 0.0;
            flow_x(sb_before);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KC; ++s) b[s] = fma(-with_sign_bit(2.0, sx, s), pc[s], b[s]);   // (P v)_i v_i - 2 v_m v_i P_im  (2 P_im is exact)
#pragma unroll
            for (int s = 0; s < KC; ++s) if (s == ms && lane == ml) b[s] = -b[s];   // ... and coordinate m's own sign
            __builtin_amdgcn_sched_barrier(0);
            store_rows(k);
        } else {
            if (lane == 0) atomicAdd(&p.status[1], 1);
            t = uni(t + tp + ts);
            ts = 0.0;
            flow_x(sb);
            store_rows(k);
        }
        errored = 0; rejected = 0; hits = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) ev[q] = 0;
        it = 0;
        ++k;
    }
    {   // state for the next segment (x flowed to the chain's current time)
        const Row<double> xo(p.In + (size_t)c * d), vo(p.In + ((size_t)p.C + (size_t)c) * d);
#pragma unroll
        for (int s = 0; s < KC; ++s)
            on_slot<KC>(s, d, dq, lane, [&] { *xo.ptr(s, lane) = x[s * WAVE] + flip_by_bit(el, sb, s); *vo.ptr(s, lane) = with_sign_bit(1.0, sb, s); });
    }
    if (lane == 0) {
        cs.t = t; cs.horizon = horizon; cs.rng_n = rng.n; cs.dead = dead ? 1 : 0;
        cs.pad = (UNTIL && !(t < p.t_stop)) ? 1 : 0;
        if (UNTIL && p.count && (cs.pad || k >= p.n_sk)) p.count[j] = k;
        p.st[c] = cs;
        if (dead) atomicAdd(&p.status[0], 1);
    }
}


