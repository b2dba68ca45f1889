#!/bin/bash
# Lab builds (not product) of conv.hip + tools/lab/conv_movers_mfma16.patch -> jspsr_amd/lib_lab/libjspsr_conv_<name>.so,
# loaded through JSPSR_LAB_LIB.  The patch (round 3, measured slower / too small to carry: profiles/r03_conv_patch_movers_mfma16_lab.txt)
# adds to conv_patch_kernel<bf16,128,128>: two MOVER waves that bring the weight stages in by LDS-DMA (JSPSR_CONV_MOVERS=1/0), and
#   movers=""                 the patched kernel as is
#   nodma="-DCONVLAB_NODMA"   mover waves run but move nothing (WRONG results): what the weight DMA itself costs
#   lb3="-DCONVLAB_LB=3"      the plain patch kernel held to the mover build's register budget (168 VGPRs)
#   mfma16="-DCONVLAB_MFMA16" every 32x32x16 product replaced by two 16x16x32 ones on the same operands (WRONG results; timing only)
# and from the product source as it is (no patch):
#   stamps="-DCONVLAB_STAMPS=1|2|3"  cycle stamps of the patch kernel's waves (tools/lab/conv_stamps.py): 1 = prologue / main loop /
#                             epilogue, 2 = per-stage phases (slows the kernel), 3 = the prologue and the epilogue in pieces
#   prio8="-DCONVLAB_PRIO=8"  workgroups with bit 8 of their index set run their main loop at s_setprio 1
set -e
cd "$(dirname "$0")/../../jspsr_amd/csrc"
make -s
mkdir -p ../lib_lab
trap 'rm -f _convlab.hip' EXIT                   # (includes are relative to csrc/)
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}; src=_convlab.hip
  case "$name" in
    stamps*|prio*) src=conv.hip;;
    *) [ -f _convlab.hip ] || { git show 0dfdb01:jspsr_amd/csrc/conv.hip > _convbase.hip;      # the patch is against that revision's kernel
                                patch -s -o ./_convlab.hip _convbase.hip ../../tools/lab/conv_movers_mfma16.patch; rm -f _convbase.hip; };;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $src -o /tmp/convlab_$name.o
  objs=$(ls _obj/*.o | grep -v "_obj/conv.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_lab/libjspsr_conv_$name.so /tmp/convlab_$name.o $objs
done
ls -la ../lib_lab
