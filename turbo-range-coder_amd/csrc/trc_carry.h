// trc_carry.h -- the append-only carry scheme of the range-coder encoder, free of any HIP dependency so that
// the very same code is unit-tested on the host (tests/test_carry_host.py) against the reference's
// behaviour: write every word at once and, on a carry, walk back adding 1 (turborc_.h:103 `_rccarry_`).
//
// The encoder logically emits a sequence of events (cy, W): "a carry reaches the words emitted so far"
// (cy) followed by "append the 32-bit word W".  Because output leaves the chip append-only, the last word
// is held back (`cache`) together with a count of 0xFFFFFFFF words behind it (`npend`): a carry turns
// cache, FF.. into cache+1, 00..; words are released to the sink only when no later carry can reach them.
// A word receives at most one carry in its lifetime (what is still to be added to the code value is
// always smaller than the range at the time the word was emitted), so after a carry everything held is final.
#pragma once
#include <stdint.h>
#ifndef TRC_HD
#if defined(__HIPCC__) || defined(__CUDACC__)
#define TRC_HD __host__ __device__ __forceinline__
#else
#define TRC_HD inline
#endif
#endif

// FF = the all-ones word of the coder's I/O width (0xFFFFFFFF: 32-bit words, 0xFFFF: 16-bit words)
template <uint32_t FF>
struct TrcCarryT {
    uint32_t cache, npend, nwords;   // nwords counts like the reference's output pointer (held-back words included)
    bool have;
    TRC_HD void start() { cache = 0; npend = 0; nwords = 0; have = false; }
    // SINK needs put32(uint32_t) and put32_slow(uint32_t) (the latter may be called in long runs)
    template <class SINK>
    TRC_HD void emit(SINK &so, bool cy, uint32_t W)
    {
        nwords++;
        if (have && !cy && npend == 0 && W != FF) {                    // the common case
            so.put32(cache); cache = W;
            return;
        }
        if (cy) {                                                       // cache+1, then zeros: all final
            so.put32((cache + 1u) & FF);
            for (; npend; npend--) so.put32_slow(0u);
            have = false;
        }
        if (!have) { cache = W; have = true; }
        else if (W == FF) npend++;
        else {
            so.put32(cache);
            for (; npend; npend--) so.put32_slow(FF);
            cache = W;
        }
    }
    // Predicated form for branch-free symbol loops: the event happens only where `on`.  SINK additionally needs
    // put32_if(bool, uint32_t).  The common cases cost no branch -- a plain append AND a carry into a held word with no
    // all-ones words behind it (release cache + 1 instead of cache: same as `emit` with npend == 0); only an all-ones
    // word, held all-ones words or the very first word take the general path.  Carries are not rare (the adaptive
    // coders see one on 10-30 % of their words), and with 64 lanes per wave a branchy carry path ran at almost every
    // renormalisation point of the wave: round 2 measured rcs encode 2.59 -> see profiles/r02_notes.md.
    template <class SINK>
    TRC_HD void emit_if(SINK &so, bool on, bool cy, uint32_t W)
    {
        const bool fast = on && have && npend == 0 && W != FF;
        so.put32_if(fast, (cache + (cy ? 1u : 0u)) & FF);
        cache = fast ? W : cache;
        nwords += fast ? 1u : 0u;
        if (on && !fast) emit(so, cy, W);
    }
    template <class SINK>
    TRC_HD void release(SINK &so)                                      // end of stream: nothing can carry any more
    {
        if (have) {
            so.put32(cache);
            for (; npend; npend--) so.put32_slow(FF);
            have = false;
        }
    }
};
typedef TrcCarryT<0xffffffffu> TrcCarry;
