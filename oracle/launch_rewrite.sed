# The two textual rewrites applied (in a pipe, never on disk) to CUDA translation units before g++ sees them.
# Used by oracle/build_ref.sh (the reference's .cu files) and oracle/Makefile (simt_selftest.cu).
#
# 1. the launch syntax g++ cannot parse:
#      Kernel<<<grid, block>>>(args);   ->   simt::Launcher(grid, block).name("Kernel").run([&]{ Kernel(args); });
s/\b\([A-Za-z_][A-Za-z0-9_]*\)[ ]*<<<\(.*\)>>>[ ]*(\(.*\));/simt::Launcher(\2).name("\1").run([\&]{ \1(\3); });/
# 2. `int angi = angf;` (descriptor angle bin, cudaSiftD.cu:353 and its dead siblings) sees NaN whenever
#    FastAtan2(0,0) = 0/0 was sampled (SURVEY Appendix B #7).  The GPU's cvt.rzi.s32.f32 turns NaN into 0 and the kernel
#    goes on to write a NaN descriptor; x86's cvttss2si yields INT_MIN and the emulation would index far outside the
#    histogram.  simt_f2i_rz() is the GPU's conversion.
s/int angi = angf;/int angi = simt_f2i_rz(angf);/
