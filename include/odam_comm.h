/*
 * odam_comm.h -- the collective of the hot path: all-gather of per-frame detection blocks before association (C ABI).
 *
 * Replaces the reference's pickle-based variable-size gather (likojack/ODAM src/utils/misc.py:90-130: all_gather of
 * torch.ByteTensor pickles padded to the largest rank) for the one thing the sharded path exchanges: after every rank has
 * detected its frames (src/processor.py:259-289), the fixed-shape blocks float32 [F_pad][30][15] + int32 [F_pad] of all ranks
 * are gathered so that every rank can replay the association loop (src/processor.py:306-345).  One process per GPU; the transport
 * is RCCL (ncclAllGather over xGMI), resolved at run time from the librccl the process already has mapped (PyTorch's, when the
 * host program is Python) -- this library carries no link-time dependency on it.  A rank's share and the result stay on the
 * device; nothing here synchronises the host.
 *
 * Bootstrap: rank 0 calls odam_comm_unique_id and hands the 128 bytes to the other ranks by whatever channel the host program
 * has (odam_amd/parallel.py broadcasts them through torch.distributed); then every rank calls odam_comm_create on its own device.
 * Return codes as in odam_sq.h (0 = OK; odam_last_error() has the message, including RCCL's own error string).
 */
#ifndef ODAM_COMM_H
#define ODAM_COMM_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct odam_comm odam_comm;

/* id [host][128]: ncclGetUniqueId */
int odam_comm_unique_id(unsigned char* id);
/* ncclCommInitRank on the calling thread's current HIP device; collective over all `world` ranks */
int odam_comm_create(const unsigned char* id, int world, int rank, odam_comm** out);
int odam_comm_destroy(odam_comm* c);
int odam_comm_info(odam_comm* c, int* world, int* rank);

/* local [dev][F_pad][30][15] float32 and n_local [dev][F_pad] int32: this rank's frames (F_pad = the padded shard length, equal
 * on all ranks); all [dev][world][F_pad][30][15], n_all [dev][world][F_pad]: rank r's block at index r.  Two ncclAllGather
 * calls in one group on `stream`; stream-ordered. */
int odam_allgather_detections(odam_comm* c, const float* local, const int* n_local, int F_pad, float* all, int* n_all, void* stream);
/* the second collective of the path (fitted parameters + boxes of the objects a rank owns): n floats per rank */
int odam_allgather_f32(odam_comm* c, const float* local, long long n, float* all, void* stream);

#ifdef __cplusplus
}
#endif
#endif
