python tools/quality.py --shape gowalla --batch 1 --epochs 3 --limit 8000 | tail -3
for B in 256 1024 4096 12500; do for cap in 4 16 64; do
python tools/quality.py --shape gowalla --batch $B --cap $cap --epochs 400 --seconds 3 --eval-every 1000 | tail -1
done; done
