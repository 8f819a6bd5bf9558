for pl in 64 320 640 1088; do echo "PLEN=$pl chunk=320"; VR2_PLEN=$pl python profiles/src/attn_decode_bench.py 0 2>&1 | grep step; done
for pc in 64 128 192 320 576; do echo "PLEN=1088 chunk=$pc"; VR2_PREFIX_CHUNK=$pc python profiles/src/attn_decode_bench.py 0 2>&1 | grep step; done
echo "suffix scaling"; python profiles/src/attn_decode_bench.py 0 63 127 255 511 1023 2>&1 | grep step
