#!/bin/bash
# one bench line per coder at its default chunk and at 4096 -> gpurun_out/all_codecs.txt (copy to profiles/)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
ALL="anscdf4s rccdf8 rccdfi8 rccdfs rccdfsm rccdfs2 rcs ansb rccdf rccdfi anscdf anscdf1 rccdf4 rccdf4i anscdf4 rccdfu16 rccdfu32 rccdfv16 rccdfv32 rccdfvz16 rccdfvz32 anscdfu16 anscdfuz16 anscdfv16 anscdfvz16 anscdfv32 anscdfvz32"
{
echo "# bench.py --no-cpu --steps 5 --warmup 1 --codec X --chunk C (0 = default chunk per coder: 512; rccdfs2 1024; rcs, ansb, rccdf, rccdfi, anscdf 1536; anscdf1 4096), 100 MB, 1 x MI355X"
echo "# enc/dec = input bytes / dominant coder kernel time (two-pass coders: both passes summed); encdec = whole step"
bash scripts/gpu_codec_sweep.sh "$ALL" "0"
bash scripts/gpu_codec_sweep.sh "$ALL" "4096"
} > gpurun_out/all_codecs.txt 2>&1
cat gpurun_out/all_codecs.txt
