for lib in ascii-renderer_amd/lib/libascii_rast.so build_variants/fb8.so; do for wc in 1 3; do echo "== $lib wc=$wc C3"; AR_LIB=$PWD/$lib AR_WORK_CLASSES=$wc timeout -k 10 100 python scripts/time_variant.py 2>&1 | grep -v amdgpu; done; done
for wc in 1 3; do echo "== wc=$wc C4"; AR_COLS=3840 AR_ROWS=2160 AR_WORK_CLASSES=$wc timeout -k 10 100 python scripts/time_variant.py 2>&1 | grep -v amdgpu; done
for wc in 1 3; do echo "== wc=$wc C5"; AR_TRIS=10000000 AR_PL=1 AR_WORK_CLASSES=$wc timeout -k 10 150 python scripts/time_variant.py 2>&1 | grep -v amdgpu; done
