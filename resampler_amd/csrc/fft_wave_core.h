// fft_wave_core.h -- what the wave-per-transform FFT kernels share (fft_wave.hip: a wave per channel; fft_pair.hip: a wave per
// two-channel stream): the LDS access helpers, the plan type with its padded layouts, the Stockham stage and the fused /
// plain first passes.  Included inside `namespace rsmp { namespace { ... } }` of a .hip file, after fft_butterflies_pk.h
// and after RSMP_FEAT is defined (the A/B switches of the slope experiments).
#pragma once

// Lanes of a wave exchange data through the wave's LDS buffer without any barrier: the hardware executes a
// wave's LDS operations in issue order.  The COMPILER, however, reasons per thread and may move a thread's
// store above its own loads of provably different addresses -- which are other lanes' data here (it did,
// in the radix-4 stage).  This pins the program order of memory operations; it emits no instruction.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// Stream pointers come out of a descriptor in memory, so the compiler knows no address space for them and
// emits FLAT loads and stores -- which count on the LDS counter too, and so tie every wait for an LDS read to
// the block's output stores.  Naming the global address space gives global_load / global_store.
typedef __attribute__((address_space(1))) float GFloat;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f4 GFloat4;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) f2 GFloat2;
__device__ __forceinline__ const GFloat* as_global(const float* p) { return (const GFloat*)p; }
__device__ __forceinline__ GFloat* as_global(float* p) { return (GFloat*)p; }

// One LDS value by ds_read_b64, which the LDS serves at 256 B/clk.  Left to itself the compiler pairs
// neighbouring loads into ds_read2_b64 / ds_read2st64_b64, which run at HALF that rate (8 LDS cycles for the
// 16 bytes per lane against 2 + 2, MI355X_MICROARCH.md LDS table) in a kernel whose bound is the LDS; a
// volatile access is never merged (and must name the LDS address space: address-space inference skips
// volatile accesses, which would otherwise become flat loads).
__device__ __forceinline__ cf lds_ld(const cf* p) {
    typedef const volatile __attribute__((address_space(3))) cf* LdsPtr;
#if RSMP_FEAT & 16   // (slope experiment: every read issued twice)
    { const cf dup = *(LdsPtr)(p); asm volatile("" :: "v"(dup)); }
#endif
    return *(LdsPtr)(p);
}
// Likewise one ds_write_b64 per value: the ds_write2_b64 the compiler forms of two costs 13 LDS cycles against 6 + 6.
__device__ __forceinline__ void lds_st(cf* p, cf v) {
#ifdef RSMP_FFT_WAVE_MERGED_STORES
    *p = v;
#else
    typedef volatile __attribute__((address_space(3))) cf* LdsPtr;
    *(LdsPtr)(p) = v;
#if RSMP_FEAT & 8   // (slope experiment: every store issued twice)
    *(LdsPtr)(p) = v;
#endif
#endif
}

// A transform of N complex points in `Rs...` Stockham stages (2 .. 4 of them), as the reference's planner orders
// them (src/fft/optimizer.rs).  Where the first two radices multiply to at most 21 values per unit (and a third
// stage exists) they run as one register pass (wave_fused_first); every later stage but the inverse's last is a
// wave_stage; the twiddle tables of all stages sit in LDS.
// LDS stores go 16 lanes at a time over 32 banks (MI355X_MICROARCH.md, LDS table; tools/fft_bank_model.py counts the
// array cycles of every pass of a plan pair).  A stage's lane i stores its value q at R (i - k) + k + q stride
// (k = i mod stride): lanes 16 apart in i are in different blocks of `stride` columns unless stride >= 16, and a
// block is (R - 1) stride values further than the lane index says -- two values per 16 lanes of shift keep the
// 16 lanes of a store on distinct banks iff (R - 1) stride + pad is a multiple of 16 values.  (Radix 7, stride 21:
// 147-value blocks, 2 values of padding; radix 8, stride 20: 4.)
constexpr int stage_out_pad(int r, int stride) { return (16 - ((r - 1) * stride) % 16) % 16; }
constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }
// Twiddles a stage keeps per column in LDS: all R - 1 of the row, or -- radix 7 and 8 -- only w, w^2 and w^4 (the
// stage multiplies the others out, see twiddle_expand; the tables of the 1176 <-> 1280 pair shrink from 39 to 29 KB).
#if !defined(RSMP_FFT_WAVE_EXACT) && !defined(RSMP_FFT_WAVE_ALL_TWIDDLES)
constexpr int fetch_count(int r) { return (r == 7 || r == 8) ? 3 : r - 1; }
#else
constexpr int fetch_count(int r) { return r - 1; }
#endif
template <int N_, int... Rs>
struct WavePlan {
    static constexpr int N = N_;
    static constexpr int kStages = sizeof...(Rs);
    static constexpr int kR[sizeof...(Rs)] = {Rs...};
    static_assert(kStages >= 2 && kStages <= 5, "stages");
    static constexpr int stride(int s) { int v = 1; for (int i = 0; i < s; ++i) v *= kR[i]; return v; }
    static_assert(stride(kStages) == N_, "radices");
    static constexpr bool kFused = kStages >= 3 && kR[0] * kR[1] <= 21;
    // Stage twiddles, unique per column: stage s (s >= 1) holds stride(s) rows of R_s - 1.  In LDS the rows of a
    // wave_stage are (R - 1) | 1 values apart: lane k reads row k, and an even row length puts lanes 16 apart
    // (radix 7: six values = 12 dwords) on the same banks.  (The fused pass reads its rows by constant index.)
    static constexpr int row(int r) { return fetch_count(r) | 1; }
    static constexpr int pitch(int s) { return kFused && s == 1 ? kR[1] - 1 : row(kR[s]); }
    static constexpr int tab(int s) { int off = 0; for (int i = 1; i < s; ++i) off += stride(i) * pitch(i); return off; }   // LDS offset of stage s
    static constexpr int src(int s) { int off = 0; for (int i = 1; i < s; ++i) off += stride(i) * (kR[i] - 1); return off; }   // offset in the plan's array
    static constexpr int kTw = tab(kStages);
    static constexpr int kRc = N_ / 2 - 1;   // real <-> complex twiddles
    // Padding between passes (LDS banks).  The first pass (fused or not) writes kUnit values per lane side by side:
    // an even kUnit puts lanes 32 / gcd(2 kUnit, 32) apart on the same banks, so one value of padding follows every
    // kPadJ units (20 values per unit: every 4) where the next stage's input distance is a multiple of that period.
    // After the blocks of a later stage: stage_out_pad, where the stage that follows reads block by block.
    // in_pad(s): what stage s's input distance N / R_s grows by; in_period(s): elements between two padding values
    // inside that distance (0 = none).
    static constexpr int kUnit = kFused ? kR[0] * kR[1] : kR[0];
    static constexpr int kNext = kFused ? 2 : 1;   // the stage that reads the first pass's output
    // (Plans above 2048 points run at the 256-register cap of their wide workgroups: the padded addressing spilled
    // there -- 2352 -> 2560 points 0.80 -> 1.00 ms -- so they keep the plain layout, but for the radix-7 blocks.)
    static constexpr bool kPadded = N_ <= 2048;
    static constexpr int first_padj() {
        if (!kPadded || kUnit % 2 != 0 || kNext >= kStages) return 0;
        const int p = 32 / gcd_c(2 * kUnit, 32);
        return (N_ / kR[kNext < kStages ? kNext : 0]) % (p * kUnit) == 0 ? p : 0;
    }
    static constexpr int kPadJ = first_padj();
    static constexpr int out_pad(int s) {
        if (s < 1 || s + 1 >= kStages || (kFused && s == 1)) return 0;
        if (stride(s) >= N_ / kR[s]) return 0;   // one block
        const int p = kPadded || (kR[s] == 7 && stride(s) == 21) ? stage_out_pad(kR[s], stride(s)) : 0;
        return p != 0 && N_ / kR[s + 1] == stride(s + 1) ? p : 0;
    }
    static constexpr int in_pad(int s) {
        if (s == kNext) return kPadJ ? (N_ / kR[s]) / (kPadJ * kUnit) : 0;
        return s >= 2 ? out_pad(s - 1) : 0;
    }
    static constexpr int in_period(int s) { return s == kNext && kPadJ && N_ / kR[s] > kPadJ * kUnit ? kPadJ * kUnit : 0; }
    static constexpr int buf_values() {   // what the wave's buffer needs: the points + bin N and its neighbour (real <-> complex passes), or the widest padded layout
        int pad = kPadJ ? N_ / (kPadJ * kUnit) : 0;
        for (int s = 1; s + 1 < kStages; ++s) {
            const int p = out_pad(s) * (N_ / stride(s + 1));
            if (p > pad) pad = p;
        }
        return N_ + (pad > 2 ? pad : 2);
    }
    static constexpr int kBuf = buf_values();
    static bool matches(uint32_t n, uint32_t n_stages, const uint32_t* radix) {
        if (n != static_cast<uint32_t>(N_) || n_stages != static_cast<uint32_t>(kStages)) return false;
        for (int s = 0; s < kStages; ++s)
            if (radix[s] != static_cast<uint32_t>(kR[s])) return false;
        return true;
    }
};

// One Stockham stage in place in the wave's LDS buffer: butterfly i reads buf[i + q*M], twiddles inputs
// 1..R-1 with w[(i mod STRIDE)*(R-1) + q-1] and writes buf[R*i - (R-1)*k + q*STRIDE]
// (butterfly4/mod.rs:316-320 etc.).  Every read of the stage is issued before its first write.
// The R - 1 twiddles of a butterfly are the powers w, w^2 .. w^(R-1) of one value.  The kernel is bound by
// LDS traffic, of which the twiddle rows were a quarter: radix 7 and 8 fetch w, w^2 and w^4 and multiply
// the others out (one or two roundings more on those twiddles; -DRSMP_FFT_WAVE_EXACT fetches all of them).
// A row is FETCHED (twiddle_fetch: kFetch<R> LDS reads, issued with the stage's data reads) and EXPANDED
// when its butterfly runs.
template <int R> constexpr int kFetch = fetch_count(R);
template <int R>
__device__ __forceinline__ void twiddle_fetch(const cf* __restrict__ w, cf (&raw)[kFetch<R>]) {
    if constexpr (kFetch<R> != R - 1) {
        raw[0] = lds_ld(w);       // (the LDS row holds w, w^2, w^4)
        raw[1] = lds_ld(w + 1);
        raw[2] = lds_ld(w + 2);
    } else {
#pragma unroll
        for (int q = 0; q < R - 1; ++q) raw[q] = lds_ld(w + q);
    }
}
template <int R>
__device__ __forceinline__ void twiddle_expand(const cf (&raw)[kFetch<R>], cf (&tw)[R]) {
    if constexpr (kFetch<R> != R - 1) {
        tw[1] = raw[0];
        tw[2] = raw[1];
        tw[4] = raw[2];
        tw[3] = cf_mul(tw[1], tw[2]);
        tw[5] = cf_mul(tw[1], tw[4]);
        tw[6] = cf_mul(tw[2], tw[4]);
        if constexpr (R == 8) tw[7] = cf_mul(tw[3], tw[4]);
    } else {
#pragma unroll
        for (int q = 1; q < R; ++q) tw[q] = raw[q - 1];
    }
}

// QS: distance of a butterfly's inputs in the buffer (N / R, or more when the producer padded its rows).
// OPAD: values of padding after every R * STRIDE outputs (one block of the next stage's columns).  With 21
// columns a half wave of 32 lanes spans two blocks, and 147 values = 294 dwords put the second block's first
// columns on the first block's last banks; two values more (298 = 42 mod 64) and every half wave of the
// stage stores conflict-free.  The next stage then reads its inputs N / R' + OPAD apart (stage_out_pad).
// IPP: the producer (the first pass) left one value of padding after every IPP of the stage's inputs (0 = none).
template <int N, int R, int STRIDE, int QS = N / R, int OPAD = 0, int IPP = 0>
__device__ __forceinline__ void wave_stage(cf* buf, const cf* __restrict__ tw, int lane) {
    constexpr int M = N / R;
    constexpr int ITER = (M + 63) / 64;
    constexpr int ROW = fetch_count(R) | 1;
    // Every LDS read of the stage -- data and twiddle rows, in the order of their use -- is issued before the
    // first butterfly: the wave then waits for a read once per stage, not once per butterfly (LDS operations
    // of a wave complete in order, so butterfly 0 runs while the later reads are still in flight).
    if constexpr (STRIDE == M && QS == M && OPAD == 0 && IPP == 0 && ITER >= 4) {
        // A plan's last stage writes every value where it read it (stride = M: the butterfly's own points), so
        // butterflies need not wait for each other's reads: of a long stage (4 or 5 trips: 64-80 values and their
        // twiddles in registers at once, which the plans of 2048 points and more paid with spills) only the next
        // trip's reads are in flight while one runs.
        cf t2[2][R], raw2[2][kFetch<R>];
        auto fetch = [&](int it) {
            const int i = lane + 64 * it;
            if ((it + 1) * 64 <= M || i < M) {
#pragma unroll
                for (int q = 0; q < R; ++q) t2[it & 1][q] = lds_ld(buf + i + q * M);
                twiddle_fetch<R>(tw + i * ROW, raw2[it & 1]);
            }
        };
        fetch(0);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = lane + 64 * it;
            if (it + 1 < ITER) fetch(it + 1);
            if ((it + 1) * 64 <= M || i < M) {
                cf twr[R], o[R];
                twiddle_expand<R>(raw2[it & 1], twr);
#pragma unroll
                for (int q = 1; q < R; ++q) t2[it & 1][q] = cf_mul(twr[q], t2[it & 1][q]);
                pdft<R>(t2[it & 1], o);
#pragma unroll
                for (int q = 0; q < R; ++q) lds_st(buf + i + q * M, o[q]);
            }
        }
        lds_order();
        return;
    }
    cf t[ITER][R], raw[ITER][kFetch<R>];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
#pragma unroll
            for (int q = 0; q < R; ++q) t[it][q] = lds_ld(buf + i + (IPP ? i / (IPP ? IPP : 1) : 0) + q * QS);
            twiddle_fetch<R>(tw + (i % STRIDE) * ROW, raw[it]);
        }
    }
    lds_order();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
            const int k = i % STRIDE;
            cf twr[R];
            twiddle_expand<R>(raw[it], twr);
#pragma unroll
            for (int q = 1; q < R; ++q) t[it][q] = cf_mul(twr[q], t[it][q]);
            cf o[R];
            pdft<R>(t[it], o);
            cf* d = buf + R * i - (R - 1) * k + (OPAD ? OPAD * (i / STRIDE) : 0);
#pragma unroll
            for (int q = 0; q < R; ++q) lds_st(d + q * STRIDE, o[q]);
        }
    }
    lds_order();
}

// Stages 0 (radix RA, stride 1, no twiddles) and 1 (radix RB, stride RA, twiddles W_(RA*RB)^(k q')) of a
// transform in ONE register pass.  The three (RA) stage-1 butterflies 3j, 3j+1, 3j+2 consume exactly the
// outputs of the seven (RB) stage-0 butterflies j + M2*q': unit j therefore takes the RA*RB points
// j + M2*m (m = q' + RB*q), runs RB radix-RA butterflies, the twiddles and RA radix-RB butterflies in
// registers, and writes the contiguous outputs RA*RB*j .. RA*RB*j + RA*RB - 1 -- what the two stages
// would have left in LDS, with one LDS round trip and the stage-1 index arithmetic gone.  Same operations
// on the same values as the separate stages (the unit twiddles of column k = 0 are skipped).
// `load(index)` yields point `index` of the stage-0 input (LDS, or samples straight from HBM).
// A unit's outputs are RA*RB values apart from the next lane's; when that is even (20 values = 40 dwords)
// the 64 lanes of a store meet on 8 bank pairs, so one value of padding follows every fused_pad<>() values
// (160: lanes 8 apart move on by a bank pair) and the next stage reads its inputs fused_qs<>() apart.
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}
// NVALID: points at index >= NVALID of the stage-0 input are zero and are neither fetched nor computed with
// (the zero padding of the forward transform: resampler_fft.rs:387-388).  Butterfly q' takes the points
// j + M2 (q' + RB q): the last NZ of its RA inputs are padding for every j (pdft_tail).
// PADJ: one value of padding after every PADJ units (WavePlan::kPadJ; 0 = none).
// `prep` sees a first pass's inputs between their loads and the first butterfly: run<ITER, K, M>(values, lane) with K values
// for each of the lane's ITER units (unit lane + 64 it exists while below M); it may change them (fft_pair.hip: the block's
// two channels are scaled to a common level there).
struct NoPrep {
    template <int ITER, int K, int M> __device__ __forceinline__ void run(cf (&)[ITER][K], int) const {}
};
template <int N, int RA, int RB, int PADJ, int NVALID = N, class Load, class Prep = NoPrep>
__device__ __forceinline__ void wave_fused_first(cf* dst, const cf* __restrict__ tw1, int lane, Load load, Prep prep = Prep()) {
    constexpr int M2 = N / (RA * RB);
    constexpr int ITER = (M2 + 63) / 64;
    cf s[ITER][RB][RA];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int j = lane + 64 * it;
        if ((it + 1) * 64 <= M2 || j < M2) {
            static_for<0, RB>([&](auto qp_c) {
                static_for<0, RA>([&](auto q_c) {
                    constexpr int m = decltype(qp_c)::value + RB * decltype(q_c)::value;
                    if constexpr (M2 * m < NVALID) s[it][decltype(qp_c)::value][decltype(q_c)::value] = load(j + M2 * m);
                });
            });
        }
    }
    if constexpr (!std::is_same<Prep, NoPrep>::value) {
        static_assert(NVALID == N, "prep sees every input");
        prep.template run<ITER, RB * RA, M2>(reinterpret_cast<cf (&)[ITER][RB * RA]>(s), lane);
    }
    cf w1[RA][RB];   // (the same for every lane: broadcast reads, fetched with the data)
#pragma unroll
    for (int k = 1; k < RA; ++k)
#pragma unroll
        for (int qp = 1; qp < RB; ++qp) w1[k][qp] = lds_ld(tw1 + k * (RB - 1) + qp - 1);
    lds_order();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int j = lane + 64 * it;
        if ((it + 1) * 64 <= M2 || j < M2) {
            static_for<0, RB>([&](auto qp_c) {
                constexpr int qp = decltype(qp_c)::value;
                // inputs q with M2 (qp + RB q) >= NVALID are zero: count them from the end
                constexpr int first_zero = M2 * qp >= NVALID ? 0 : (NVALID - M2 * qp + M2 * RB - 1) / (M2 * RB);
                constexpr int NZ = first_zero >= RA ? 0 : RA - first_zero;
                cf o[RA];
                pdft_tail<RA, NZ>(s[it][qp], o);
#pragma unroll
                for (int k = 0; k < RA; ++k) s[it][qp][k] = o[k];
            });
#pragma unroll
            for (int k = 0; k < RA; ++k) {
                cf u[RB], o[RB];
                u[0] = s[it][0][k];
#pragma unroll
                for (int qp = 1; qp < RB; ++qp)
                    u[qp] = k == 0 ? s[it][qp][k] : cf_mul(w1[k][qp], s[it][qp][k]);
                pdft<RB>(u, o);
#pragma unroll
                for (int qq = 0; qq < RB; ++qq) lds_st(dst + RA * RB * j + (PADJ ? j / (PADJ ? PADJ : 1) : 0) + k + RA * qq, o[qq]);
            }
        }
    }
    lds_order();
}

// Stage 0 alone (stride 1, no twiddles) for the plans that do not fuse it with stage 1: butterfly i takes the
// points i + q N / R through `load` (LDS, or samples straight from HBM; points at index >= NVALID are zero) and
// writes R i + q.
template <int N, int R, int PADJ, int NVALID = N, class Load, class Prep = NoPrep>
__device__ __forceinline__ void wave_first(cf* dst, int lane, Load load, Prep prep = Prep()) {
    constexpr int M = N / R;
    constexpr int ITER = (M + 63) / 64;
    constexpr int first_zero = (NVALID + M - 1) / M;            // inputs q >= first_zero are zero for every i
    constexpr int NZ = first_zero >= R ? 0 : R - first_zero;
    cf t[ITER][R];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
#pragma unroll
            for (int q = 0; q < R - NZ; ++q) t[it][q] = load(i + q * M);
        }
    }
    if constexpr (!std::is_same<Prep, NoPrep>::value) {
        static_assert(NVALID == N, "prep sees every input");
        prep.template run<ITER, R, M>(t, lane);
    }
    lds_order();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
            cf o[R];
            pdft_tail<R, NZ>(t[it], o);
#pragma unroll
            for (int q = 0; q < R; ++q) lds_st(dst + R * i + (PADJ ? i / (PADJ ? PADJ : 1) : 0) + q, o[q]);
        }
    }
    lds_order();
}

