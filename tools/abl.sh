for v in "" NOMFMA NOLOAD NOLOADB NOLOADDABL_NOLOADB NOEPI; do
  if [ -z "$v" ]; then unset DIFFSEP_LIB; else export DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_$v.so; fi
  echo "== variant ${v:-BASE}"; python tools/bench_conv.py bf16 2>&1 | grep -E "k3   64->  64  256|k3  128->  64  256|k1  128->  64  256|k3  128-> 128   16x16"
done
