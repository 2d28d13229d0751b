# lab builds of the fused FFT front-end with parts compiled out (FF_ABL bits: 1 mel sums, 2 the two 16-point DFTs + twiddles, 4 quad radix-4, 8 most magnitudes, 16 sample loads)
mkdir -p lab
objs=$(ls efficient_tts_amd/build/*.o | grep -v efts_frontend.o)
for a in ${ABLS:-0 1 2 4 8 16 31}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DFF_ABL=$a -c efficient_tts_amd/csrc/efts_frontend.hip -o lab/fe_abl$a.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lab/fft_abl$a.so lab/fe_abl$a.o $objs &
done
wait
ls -la lab/fft_abl*.so
