// Host unit test of turbo-range-coder_amd/csrc/trc_carry.h (the append-only carry scheme the HIP range
// coders use) against the reference's behaviour: write each word immediately and, on a carry, ripple +1
// backwards through the words already written (turborc_.h:103).  Event streams are random but biased so
// that runs of 0xFFFFFFFF words and carries into them -- astronomically rare on real data -- happen often.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../turbo-range-coder_amd/csrc/trc_carry.h"

struct Sink {
    std::vector<uint32_t> w;
    void put32(uint32_t v) { w.push_back(v); }
    void put32_slow(uint32_t v) { w.push_back(v); }
    void put32_if(bool take, uint32_t v) { if (take) w.push_back(v); }
};

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

template <uint32_t FF>
static int run(const char *what)
{
    long cases = 0, ffruns = 0, carries_into_ff = 0;
    for (int trial = 0; trial < 20000; trial++) {
        const int nev = 1 + rnd() % 60;
        std::vector<uint32_t> ref;            // reference: immediate writes + backward ripple
        std::vector<bool> carried;            // a word may receive at most one carry (arithmetic-coding invariant)
        TrcCarryT<FF> c; c.start();
        Sink s;
        for (int i = 0; i < nev; i++) {
            uint32_t W;
            const uint32_t r = rnd() % 100;
            if (r < 35) W = FF; else if (r < 45) W = FF - 1u; else if (r < 50) W = 0; else W = rnd() & FF;
            // a carry is only possible if some word exists and the ripple would stop at a word that has not
            // carried yet (what the coder guarantees); pick carries that respect the invariant
            bool cy = false;
            if (!ref.empty() && rnd() % 4 == 0) {
                size_t j = ref.size();
                bool ok = true;
                do { j--; if (carried[j]) { ok = false; break; } } while (ref[j] == FF && j > 0);
                if (ok && !(ref[j] == FF)) cy = true;          // ripple ends inside the buffer on a fresh word
            }
            if (cy) {
                size_t j = ref.size();
                do { j--; ref[j] = (ref[j] + 1) & FF; carried[j] = true; } while (ref[j] == 0 && j > 0);
                if (j + 1 < ref.size()) carries_into_ff++;
            }
            ref.push_back(W); carried.push_back(false);
            // odd trials go through the predicated entry point, mixed with events that are switched off
            if (trial & 1) { if (rnd() % 3 == 0) c.emit_if(s, false, rnd() & 1, rnd() & FF); c.emit_if(s, true, cy, W); }
            else c.emit(s, cy, W);
            if (W == FF) ffruns++;
            if (c.nwords != ref.size()) { printf("nwords mismatch\n"); return 1; }
        }
        c.release(s);
        if (s.w != ref) {
            printf("MISMATCH in trial %d (%zu vs %zu words)\n", trial, s.w.size(), ref.size());
            return 1;
        }
        cases++;
    }
    printf("ok: %s: %ld event streams, %ld all-ones words, %ld carries rippling through all-ones runs\n", what, cases, ffruns, carries_into_ff);
    return (ffruns > 1000 && carries_into_ff > 1000) ? 0 : 2;
}

int main()
{
    int rc = run<0xffffffffu>("32-bit words");
    if (!rc) rc = run<0xffffu>("16-bit words");
    return rc;
}
