/*
 * odam_config.h -- the library's process-wide switches (C ABI).
 *
 * The reference has no counterpart: its kernels are chosen by cuDNN / cuBLAS heuristics inside PyTorch
 * (src/models/backbone.py:59-94, src/models/transformer.py:132-238 run through torch.nn.functional).  Here every choice that
 * changes which kernel runs -- and with it the order of an fp32 summation -- is an explicit, named switch of the process, set
 * by the host program; the library never reads the environment for them.  Keys, defaults and ranges: odam_amd/csrc/odam_config.h.
 *   cg.ring cg.f32 cg.fuse cg.fuse_bf16 cg.s1 cg.ut cg.tiles cg.force cg.presplit cg.mfma16 att.x3 att.bf16_mfma stem.rows
 *   assoc.persist sq.split sq.split_wait sq.lanes stem.pool cg.pin cg.small_x3 assoc.sk_fast assoc.hungarian assoc.merge sq.two_per_cu
 * Return codes as in odam_sq.h (0 = OK; odam_last_error() has the message).  Setting a switch affects launches made after
 * the call; handles created earlier keep what they sized from it (assoc.persist is read when a handle is finalised).
 */
#ifndef ODAM_CONFIG_H
#define ODAM_CONFIG_H
#ifdef __cplusplus
extern "C" {
#endif

int odam_config_set(const char* key, int value);
int odam_config_get(const char* key, int* value);
/* buf [host][n] receives "key=value key=value ..." for the whole table; a value that differs from its default carries a '*' */
int odam_config_describe(char* buf, int n);

#ifdef __cplusplus
}
#endif
#endif
