#!/bin/bash
# link_reference_harness.sh -- BUILD CONTAINER ONLY (needs /root/reference; nothing of it enters this repo).
#
# Proves the drop-in claim of INTEGRATION.md section 2: the reference's own bench harness (turborc.c, bench()
# turborc.c:420-579) links UNCHANGED against libturborc_hip.so and its hot ids -- 1, 42-47, 50, 52, 53, 56-58,
# 60-66 -- bind to the library, not to the reference's CPU objects.
#
# Recipe (what a reference maintainer would put behind `make HIP=1`):
#   1. compile the reference objects as its makefile does (makefile:118,181,195-201,270-275), sources read where
#      they lie, objects written to $OUT (default /tmp/trc_link);
#   2. in rccdf.o / rc_s.o / anscdfs.o / anscdfx.o LOCALIZE every global the library exports
#      (objcopy --localize-symbols): the objects keep their non-hot functions (rc4senc, rccdfenc8, mbc_c, ...),
#      which the harness still needs at link time, but no longer define the hot names;
#   3. link turborc.o + objects + -lturborc_hip;
#   4. assert with nm that every hot symbol is UNDEFINED in the executable and (readelf) NEEDED from the library,
#      and with LD_DEBUG=bindings (when a GPU-less dlopen of the library works) that the dynamic linker binds
#      them to libturborc_hip.so.
# turborc.c includes libsais' header unconditionally unless _BWTDIV is defined (turborc.c:58-63); the libsais
# submodule is empty in this checkout, so the harness is compiled with -D_BWTDIV (headers of the vendored
# libdivsufsort) and WITHOUT -D_BWT: no BWT code is compiled or linked, no stand-in file is written.
set -euo pipefail
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/trc_link}
LIBDIR=$ROOT/turbo-range-coder_amd
[ -f "$REF/turborc.c" ] || { echo "no reference at $REF (this script runs in the build container only)"; exit 2; }
[ -f "$LIBDIR/libturborc_hip.so" ] || make -s -C "$LIBDIR"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$OUT"

CF="-O3 -w -DNDEBUG -D_ANS -D_TRANSPOSE -D_NCPUISA -D_BWTDIV -mavx -mpopcnt -I$REF"
LIBSRC="rc_ss rc_s rccdf rcutil bec_b rccm_s rccm_ss rcqlfc_s rcqlfc_ss rcqlfc_sf cpu transpose transpose_"
for f in $LIBSRC turborc; do gcc $CF -c "$REF/$f.c" -o "$f.o" & done
gcc $CF -falign-loops=32 -c "$REF/anscdf.c" -o anscdfs.o &
gcc ${CF/-mavx -mpopcnt/-march=haswell} -falign-loops=32 -c "$REF/anscdf.c" -o anscdfx.o &
gcc ${CF/-mavx -mpopcnt/-march=haswell} -c "$REF/transpose.c" -o transpose_avx2.o &
wait

# ---- the unmodified reference tool (all reference objects, no library): the other side of the file-format interop test
# (tests/test_gpu_parity.py::test_reference_file_format_interop); CPU only
OBJS0=""
for f in $LIBSRC anscdfs anscdfx transpose_avx2 turborc; do OBJS0="$OBJS0 $f.o"; done
gcc $OBJS0 -lrt -lpthread -lm -o turborc_ref

# ---- 2. hot symbols = (globals the four hot-path objects define) x (what the library exports) ----------------
nm -D --defined-only "$LIBDIR/libturborc_hip.so" | awk '$2 ~ /^[TDB]$/ {print $3}' | sort -u > lib_exports.txt
: > hot.txt
for o in rccdf.o rc_s.o anscdfs.o anscdfx.o; do
    nm --defined-only "$o" | awk '$2 ~ /^[TDBR]$/ {print $3}' | sort -u > "$o.defs"
    comm -12 "$o.defs" lib_exports.txt > "$o.hot"
    objcopy --localize-symbols="$o.hot" "$o"
    cat "$o.hot" >> hot.txt
done
sort -u hot.txt -o hot.txt
echo "hot symbols taken from the library: $(wc -l < hot.txt)"

# ---- 3. link ------------------------------------------------------------------------------------------------
OBJS=""
for f in $LIBSRC anscdfs anscdfx transpose_avx2 turborc; do OBJS="$OBJS $f.o"; done
gcc $OBJS -L"$LIBDIR" -lturborc_hip -Wl,-rpath,"$LIBDIR" -lrt -lpthread -lm -o turborc_hip

# ---- 4. assertions ------------------------------------------------------------------------------------------
fail=0
MUST="cdfini rccdfsenc rccdfsbdec rccdfsldec rccdfsvbdec rccdfsvldec rccdfsmenc rccdfsmbdec rccdfsmldec rccdfs2enc rccdfsb2dec rccdfsl2dec
rccdfenc rccdfdec rccdfienc rccdfidec rccdf4enc rccdf4dec rccdf4ienc rccdf4idec rccdfenc8 rccdfdec8 rccdfienc8 rccdfidec8 rcsenc rcsdec
rccdfuenc16 rccdfudec16 rccdfvenc16 rccdfvdec16 rccdfvzenc16 rccdfvzdec16 rccdfuenc32 rccdfvenc32 rccdfvzenc32
anscdfenc anscdfdec anscdfencs anscdfdecs anscdfencx anscdfdecx anscdf4enc anscdf4dec anscdf1enc anscdf1dec anscdf4senc anscdf4sdec
anscdfuenc16 anscdfuzenc16 anscdfvenc16 anscdfvzenc16 anscdfvenc32 anscdfvzenc32 ansbc ansbd"
nm turborc_hip > exe.nm
for s in $MUST; do
    t=$(awk -v s="$s" '$NF == s {print $(NF-1)}' exe.nm | sort -u | tr -d '\n')
    # "U" = imported; a lower-case "t"/"d" beside it is the localized reference copy, which nothing outside its own
    # object can reach.  Any upper-case definition (T/D/B/R) would pre-empt the library: that is the failure to catch.
    case "$t" in
        *[TDBRW]*) echo "FAIL: $s is globally defined ('$t') in the executable: the CPU copy would pre-empt the library"; fail=1 ;;
        *) ;;
    esac
done
nu=$(for s in $MUST; do awk -v s="$s" '$NF == s && $(NF-1) == "U"' exe.nm; done | wc -l)
echo "hot symbols undefined in the executable (bound at load time): $nu"
[ "$nu" -ge 40 ] || { echo "FAIL: only $nu hot symbols are imported"; fail=1; }
readelf -d turborc_hip | grep -q 'libturborc_hip.so' || { echo "FAIL: libturborc_hip.so is not NEEDED"; fail=1; }
# no hot symbol may be exported by the executable's own dynamic table either (it would pre-empt the library)
if nm -D --defined-only turborc_hip 2>/dev/null | awk '{print $3}' | grep -Fxf hot.txt; then echo "FAIL: executable exports hot symbols"; fail=1; fi

# dynamic binding as the loader sees it (works without a GPU: no call is made, the program prints its usage and exits)
if LD_DEBUG=bindings ./turborc_hip > /dev/null 2> ld.txt < /dev/null; then :; fi
LD_BIND_NOW=1 LD_DEBUG=bindings ./turborc_hip > /dev/null 2> ld.txt < /dev/null || true
for s in rccdfs2enc rccdfsb2dec anscdfenc anscdf4senc rcsenc rccdfenc cdfini; do
    if grep -q "binding file .*turborc_hip \[0\] to .*libturborc_hip.so \[0\]: normal symbol \`$s'" ld.txt; then :;
    else echo "FAIL: loader did not bind $s to libturborc_hip.so"; fail=1; fi
done
if grep "normal symbol" ld.txt | grep -F -f <(sed 's/.*/`&'"'"'/' hot.txt) | grep -v "to .*libturborc_hip.so" | grep -v "binding file .*libturborc_hip.so" | head -5 | grep .; then
    echo "FAIL: a hot symbol is bound elsewhere"; fail=1
fi
# --install: keep the linked harness next to the reference oracle build (oracle/_ref/ is git-ignored but travels to the
# GPU box with the gpurun snapshot, like libtrc_ref.so), so that tests/test_gpu_parity.py can run the REFERENCE's own
# bench() against the GPU library there.
if [ $fail -eq 0 ] && [ "${1:-}" = "--install" ]; then
    mkdir -p "$ROOT/oracle/_ref"
    gcc $OBJS -L"$LIBDIR" -lturborc_hip -Wl,-rpath,'$ORIGIN/../../turbo-range-coder_amd' -lrt -lpthread -lm -o "$ROOT/oracle/_ref/turborc_hip"
    cp turborc_ref "$ROOT/oracle/_ref/turborc_ref"
    echo "installed oracle/_ref/turborc_hip and oracle/_ref/turborc_ref"
fi
[ $fail -eq 0 ] && echo "OK: $OUT/turborc_hip = reference turborc.c + reference non-hot objects + libturborc_hip.so; hot ids bind to the library"
exit $fail
