/* nastar_dev_flags.h -- A/B switches of the DEVELOPMENT build of libnastar_hip (make -C neural-astar_amd/csrc dev ->
 * lib/libnastar_hip_dev.so, compiled with -DNASTAR_DEV).  They select older instruction streams of the search step so that the
 * stream-equality tests (tests/test_gpu_parity.py: every stream gives identical histories, paths, step counts and selection logs) and the
 * probes under tools/ can still compare them; the product library (include/nastar.h) rejects these bits with NASTAR_ERR_UNSUPPORTED. */
#ifndef NASTAR_DEV_FLAGS_H_
#define NASTAR_DEV_FLAGS_H_
#define NASTAR_FLAG_NO_ASM 8    /* forward / backward: the compiler-generated step instead of the hand-scheduled instruction stream */
#define NASTAR_FLAG_ASM_V2 16   /* forward: the round-2 stream (ord keys; what maps with a negative cost take anyway) where the round-4 one applies */
#define NASTAR_FLAG_NO_DIVE 32  /* forward, 64x64 maps: the hand-scheduled stream without its "dive" fast path */
#define NASTAR_FLAG_ASM_V3 128  /* forward: the round-3 stream where the round-4 one applies */
#endif
