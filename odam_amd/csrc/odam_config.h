// odam_config.h -- the library's process-wide switches in ONE table (include/odam_config.h is the C ABI over it).
// The library never reads the environment for them: a switch that changes which kernel runs -- and with it the order of an
// fp32 summation -- is set through odam_config_set by the host program, which can log what it set (bench.py prints the
// table in its JSON line).  Defaults below are what every measurement in DESIGN.md was taken with.
#pragma once

namespace odam_cfg {
enum Key {
    CG_RING,        // 256-row LDS-DMA ring kernel: 0 never, 1 for layers that fill the device (default), 2 whenever eligible
    CG_F32,         // fp32 layers on the ring kernel: 0 = fp32 matrix instruction on 128x128 tiles, 2 = exact three-way bf16 split (default)
    CG_FUSE,        // fp32 bottlenecks on the tile: 0 off, 1 expand only, 2 with the chained reduce (default)
    CG_FUSE_BF16,   // bf16 bottlenecks on the tile: same values
    CG_S1,          // bf16 3x3 stride 1: one LDS window per (channel slice, ky) serves the three horizontal taps (default 1)
    CG_UT,          // uniform-tap LDS-DMA gather of the 128x128 / 128x64 / 64x64 tiles (default 1; 0 = register-staged gather)
    CG_TILES,       // bit 0: 8-wave 128x128 tiles, bit 1: 8-wave 128x64, bit 2: four-stage 64x64, bit 3: 16-wave 512x64 ring tiles for
                    // layers of <= 64 output channels, bit 4: 16 waves of 64x64 on the 256x256 tiles of the plain bf16 layers (default 31)
    CG_FORCE,       // tests: pin one small tile shape (1: 128x64, 2: 64x64, 3: 4-wave 128x128; default 0)
    CG_PRESPLIT,    // fp32 split mode takes pre-split filters where the caller supplies them (default 1)
    CG_MFMA16,      // fp32 split mode on v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on it) instead of 32x32x16: 0 off,
                    // 1 plain layers, 2 + the 128-column bottleneck launches, 3 (default) + the 64-column ones
    ATT_X3,         // fp32 attention through the exact bf16 split (default 1; 0 = fp32 matrix instruction)
    ATT_BF16_MFMA,  // bf16 attention on the bf16 matrix instruction (default 1)
    STEM_ROWS,      // conv1 as a 7x1 row convolution on the ring kernel where it fills the device (default 1)
    ASSOC_PERSIST,  // association: matching GNN as one persistent launch where the residency gate allows: 2 (default) rows dealt to the XCDs, one
                    // device-wide barrier per layer and XCD-local ones for the rest (gnn_rowpart_kernel); 1 five device-wide barriers per layer
                    // (gnn_persistent_kernel; bit-identical); 0 the launch sequence
    SQ_SPLIT,       // super-quadric fit: workgroups per object at most (default 32)
    SQ_SPLIT_WAIT,  // ... bounded wait for a partner's rows, 100 MHz ticks (default 3000; 0 = never wait)
    SQ_LANES,       // ... lanes per view in the extent search (default 0 = chosen by the cost model)
    STEM_POOL,      // conv1's 3x3 / stride 2 max-pool on the tile of the row-convolution stem (fp32 split mode, sixteen-wave tiles): default 1
    CG_PIN,         // 1: the kernel (tile shape, matrix instruction, fusion) chosen for a layer does not depend on how many rows it has --
                    // every size-dependent choice is taken as for a problem that fills the device, so a frame's result is bit-identical
                    // whatever batch or shard it arrives in (odam_amd/pipeline.py; slower for small batches).  Default 0
    CG_SMALL_X3,    // fp32 split mode (cg.f32 = 2) on the small tiles too -- the layers the ring kernel does not take (few rows, decoder, heads):
                    // both operands split in registers, six products on the bf16 matrix instruction.  Default 1; 0 = v_mfma_f32_32x32x2_f32 there
    ASSOC_SK_FAST,  // Sinkhorn on one wavefront (<= 128 rows, <= 32 columns; assoc.hip sinkhorn_wave_kernel): 1 (default) scaling factors absorbed
                    // into the potentials every 10 iterations, restarting with 1 and then in log space if a factor leaves its range;
                    // 2 / 3 = start at that rung (tests); 0 = the sixteen-wave log-space kernel
    ASSOC_HUNGARIAN,  // the Hungarian step of the association (associator.py:19-35): 1 on the device behind the Sinkhorn loop (odam_assoc_hungarian: scipy's
                    // solver restated on one wavefront), 0 (default) scipy on the host as the reference does -- measured 0.01 ms per frame FASTER than the
                    // device form (the host waits for the device either way).  Read by the HOST side (odam_amd/associator.py), the library only keeps it
    ASSOC_MERGE,    // association: the attention's merge projection folded into the first MLP layer at load time (W0' = [W0x | W0m Wm], b0' = b0 + W0m bm,
                    // products in binary64, rounded once): one launch per fuser layer and one barrier stage per matching layer fewer.  The same
                    // function of the weights, not the same roundings (differences of ~1e-7 relative in the layer's hidden row).  Read when a
                    // handle is finalised.  Default 1: every reference-run fixture holds in both forms (assignment probabilities within 2e-4, identical
                    // decisions through the whole chain), the association loop is 0.045 ms per frame shorter (0.59 -> 0.54 at 42 tracks)
    SQ_TWO_PER_CU,  // super-quadric fit, launches of more objects than CUs (one workgroup per object, no view split): 1 (default) = read the view counts
                    // back and either fit the objects longest first (unequal view counts: the launch ends with the last long object) or, for about
                    // equal ones, run 512-thread workgroups two per CU (500 x 256: 29 -> 24 ms per launch, 300 x 128 -17 %).  Same bits.  0 = neither
    N_KEYS
};
int get(Key k);
void set(Key k, int v);
}  // namespace odam_cfg
