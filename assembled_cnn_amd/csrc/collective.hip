// asm_allreduce_bucket: the gradient exchange of one bucket for a caller that owns its RCCL communicator.
//
// The reference's only collective is MirroredStrategy's gradient all-reduce (official/utils/misc/distribution_utils.py:24-45:
// every trainable variable's gradient summed over the replicas).  The Python host of this repository leaves the communicator to
// torch.distributed (dp.GradSync -> ProcessGroupNCCL, which IS RCCL over xGMI on ROCm); a C caller
// without PyTorch has its own ncclComm_t and gets the same bucket semantics here: an in-place SUM over `count` elements of
// the flat gradient arena on the communicator's stream, ordered behind everything the producer stream has enqueued so far
// (the backward kernels that wrote the bucket).  The caller joins the communicator's stream back into its consumer stream
// (asm_stream_join) before the optimiser; the 1/N is folded into asm_sgd_momentum's grad_scale, as in the Python host.
//
// librccl is resolved at first use (dlopen), so libasm_hip.so carries no link-time dependency on it: a single-GPU caller
// never loads it.
#include "common.h"

#include <dlfcn.h>

#include <mutex>

namespace {

// rccl.h: ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*error_string_fn)(int);
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9;   // rccl.h: ncclRedOp_t / ncclDataType_t

std::mutex g_mu;
all_reduce_fn g_all_reduce = nullptr;
error_string_fn g_error_string = nullptr;
bool g_tried = false;

char g_dl_error[256] = "";

bool resolve() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_tried) return g_all_reduce != nullptr;
  g_tried = true;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    const char* e = dlerror();               // (reading it clears it: read once, keep the text)
    snprintf(g_dl_error, sizeof(g_dl_error), "%s", e ? e : "dlopen failed");
    return false;
  }
  g_all_reduce = reinterpret_cast<all_reduce_fn>(dlsym(h, "ncclAllReduce"));
  g_error_string = reinterpret_cast<error_string_fn>(dlsym(h, "ncclGetErrorString"));
  if (!g_all_reduce) snprintf(g_dl_error, sizeof(g_dl_error), "no ncclAllReduce in librccl");
  return g_all_reduce != nullptr;
}

}  // namespace

extern "C" int asm_allreduce_bucket(void* buf, size_t count, int dtype, void* nccl_comm, void* comm_stream,
                                    void* producer_stream) {
  ASM_REQUIRE(buf && nccl_comm, "allreduce_bucket: null buffer or communicator");
  ASM_REQUIRE(comm_stream != nullptr, "allreduce_bucket: the exchange needs a stream of its own (not the null stream)");
  if (dtype != ASM_F32 && dtype != ASM_BF16) ASM_FAIL(ASM_ENOTSUP, "allreduce_bucket: dtype %d (float32 or bfloat16 buckets)", dtype);
  if (count == 0) return ASM_OK;
  // the exchange is issued, not recorded: inside asm_tape_begin .. asm_tape_end the join below would become a JOIN node of a
  // tape that never holds the all-reduce itself
  if (asm_tape_on) ASM_FAIL(ASM_EINVAL, "allreduce_bucket: not while a launch tape is being recorded (split the tape at the bucket)");
  if (!resolve()) ASM_FAIL(ASM_ENOTSUP, "allreduce_bucket: librccl.so not usable (%s)", g_dl_error);
  if (producer_stream != comm_stream) {
    if (int rc = asm_stream_join(comm_stream, producer_stream)) return rc;   // the bucket's gradients are final before the exchange reads them
  }
  const int rc = g_all_reduce(buf, buf, count, dtype == ASM_F32 ? kNcclFloat32 : kNcclBfloat16, kNcclSum, nccl_comm,
                              (hipStream_t)comm_stream);
  if (rc != 0) ASM_FAIL(ASM_EHIP, "allreduce_bucket: ncclAllReduce: %s", g_error_string ? g_error_string(rc) : "error");
  return ASM_OK;
}
