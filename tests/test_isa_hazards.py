"""The shipped ISA obeys the wait-state rule the compiler enforces on its own code.

hipcc for gfx950 keeps two wait states between a VALU write of an SGPR / VCC and a VALU read of it (LLVM GCNHazardRecognizer,
gfx940+); nothing inside an `asm` string is padded.  Rounds 3-4 shipped ~160 hand-written carry-chain sites with zero wait states
(`rcs` encode / decode, `anscdf` / `ansb` decode).  scripts/check_isa_hazards.py walks the disassembly of every kernel in the
library; this test pins "no site" -- on the library as built here (cross-compiled: no GPU needed) -- and that the scanner does
find the round-4 patterns when they are put in front of it (a scanner that finds nothing anywhere proves nothing).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_isa_hazards as H   # noqa: E402

LIB = os.path.join(ROOT, "turbo-range-coder_amd", "libturborc_hip.so")

LISTING = """
0000000000001000 <kernel_a>:
	v_sub_co_u32_e32 v24, vcc, v3, v60                         // 000000001000: 34303903
	v_subb_co_u32_e32 v61, vcc, v2, v62, vcc                   // 000000001004: 3A7A7D02
	v_subb_co_u32_e64 v63, vcc, 0, 0, vcc                      // 000000001008: D11E6A3F 01A90080
	v_bfi_b32 v3, v63, v3, v24                                 // 000000001010: D1CA0003 0462073F
0000000000002000 <kernel_b>:
	v_cmp_gt_u32_e32 vcc, 0x8000, v16                          // 000000002000: 7D9820FF 00008000
	v_perm_b32 v11, v16, v10, v59                              // 000000002008: D1ED000B 04EE1510
	v_cmp_gt_u32_e64 s[16:17], s64, v19                        // 000000002010: D0CC0010 00022640
	v_cndmask_b32_e32 v16, v16, v11, vcc                       // 000000002018: 00201710
	v_cndmask_b32_e64 v19, v19, v12, s[16:17]                  // 00000000201C: D1000013 00421913
0000000000003000 <kernel_c>:
	v_add_co_u32_e32 v1, vcc, v2, v3                           // 000000003000: 32020702
	s_nop 1                                                    // 000000003004: BF800001
	v_addc_co_u32_e32 v4, vcc, v5, v6, vcc                     // 000000003008: 38080D05
	v_readfirstlane_b32 s4, v1                                 // 00000000300C: 7E080501
	v_add_u32_e32 v7, s4, v7                                   // 000000003010: 680E0E04
0000000000004000 <kernel_d>:
	v_sub_co_u32_e32 v24, vcc, v3, v60                         // 000000004000: 34303903
	v_mad_u32_u24 v9, v8, v7, v6                               // 000000004004: D1C30009 041A0F08
	v_sub_co_u32_e64 v25, s[10:11], v4, v60                    // 00000000400C: D1190A19 00027904
	v_subb_co_u32_e32 v61, vcc, v2, v9, vcc                    // 000000004014: 3A7A1302
	v_lshrrev_b32_e32 v30, 5, v31                              // 000000004018: 203C3E85
	v_subb_co_u32_e64 v26, s[10:11], v5, v9, s[10:11]          // 00000000401C: D11E0A1A 002A1305
"""


def test_scanner_finds_the_round4_patterns():
    sites = H.scan(LISTING)
    per = {}
    for k, w, r, since, hit in sites:
        per.setdefault(k, []).append((w.split()[0], r.split()[0], since))
    # kernel_a: sub_co -> subb (0 states), subb -> subb_e64 (0 states); the first write is REPLACED by the second before the third reads
    assert per["kernel_a"] == [("v_sub_co_u32_e32", "v_subb_co_u32_e32", 0), ("v_subb_co_u32_e32", "v_subb_co_u32_e64", 0)]
    # kernel_b: the static rANS pair block -- two instructions between each compare and its select: clean
    assert "kernel_b" not in per
    # kernel_c: s_nop 1 = two states: clean; v_readfirstlane -> VALU read of that SGPR straight behind it: a site
    assert per["kernel_c"] == [("v_readfirstlane_b32", "v_add_u32_e32", 0)]
    # kernel_d: round 5's interleaved chains on two carry registers: clean
    assert "kernel_d" not in per


def test_library_has_no_valu_sgpr_hazard_site():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    text = H.disassemble(LIB)
    sites = H.scan(text)
    assert len(text) > 100000, "disassembly suspiciously short"
    assert not sites, "VALU write -> VALU read of an SGPR / VCC with < 2 wait states:\n" + "\n".join(
        "%s: %s -> %s (%d)" % (k, w, r, since) for k, w, r, since, _ in sites[:20])
