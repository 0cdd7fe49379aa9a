// temporary: entry points not implemented yet
#include "lld_common.h"
extern "C" {
int lld_local_ba(lld_ctx*, const lld_ba_window*, const lld_ba_params*, volatile const int*, lld_ba_result*) { return LLD_ERR_UNSUPPORTED; }
int lld_ba_batch_create(lld_ctx*, int, const lld_ba_window*, const lld_ba_params*, lld_ba_batch**) { return LLD_ERR_UNSUPPORTED; }
int lld_ba_batch_solve(lld_ba_batch*, volatile const int*) { return LLD_ERR_UNSUPPORTED; }
int lld_ba_batch_download(lld_ba_batch*, int, lld_ba_result*) { return LLD_ERR_UNSUPPORTED; }
int lld_ba_batch_stats(lld_ba_batch*, lld_ba_stats*) { return LLD_ERR_UNSUPPORTED; }
int lld_ba_batch_result_records(lld_ba_batch*, void**, uint64_t*) { return LLD_ERR_UNSUPPORTED; }
int lld_ba_batch_phase_ms(lld_ba_batch*, double*) { return LLD_ERR_UNSUPPORTED; }
int lld_ba_batch_kernel_stats(lld_ba_batch*, int64_t*, double*) { return LLD_ERR_UNSUPPORTED; }
void lld_ba_batch_destroy(lld_ba_batch*) {}
}
