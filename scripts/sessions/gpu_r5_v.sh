# order-1 chains: parity + kernel times
timeout 900 python -m pytest tests -m gpu -x -q -k "anscdf1 or ANSO1 or o1 or order1" 2>&1 | tail -2
timeout 300 bash scripts/gpu_kstats.sh o1main "--codec anscdf1 --no-beyond" 2>&1 | grep "walk\|sort\|place\|codeq\|dec_k"
