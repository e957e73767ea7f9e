set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --anchors 100000 --width 800 --height 800 --steps 5 --warmup 2 2>&1 | tail -20
