// Shared by the translation units that implement include/rln_amd.h: error capture at the C boundary (never let an
// exception cross it) and the handle types.
#pragma once
#include <exception>
#include <memory>
#include <string>

#include "../../include/rln_amd.h"
#include "msm.h"
#include "prover.h"

namespace rlnamd {
extern thread_local std::string g_last_error;
int fail(const std::exception& e);
void fill_prover_info(const Prover& P, rlnamd_prover_info* info);
// n verifications on host threads; throws MalformedVerifyingKey when nv does not match the key
void verify_many_common(const Zkey& zk, size_t n, const uint8_t* proofs, const uint8_t* values_le, size_t nv, int threads,
                        uint8_t* ok);
}  // namespace rlnamd

#define RLN_TRY try {
#define RLN_CATCH                        \
  return RLNAMD_OK;                      \
  }                                      \
  catch (const std::exception& e) {      \
    return rlnamd::fail(e);              \
  }                                      \
  catch (...) {                          \
    rlnamd::g_last_error = "unknown error"; \
    return RLNAMD_ERR;                   \
  }

struct rlnamd_prover {
  std::unique_ptr<rlnamd::Prover> p;
};
struct rlnamd_msm {   // one of the two (rlnamd_msm_new / rlnamd_msm_new_g2)
  std::unique_ptr<rlnamd::MsmG1> m;
  std::unique_ptr<rlnamd::MsmG2> m2;
};
// pool.cpp
rlnamd::Prover* rlnamd_pool_replica_prover(rlnamd_pool* p, size_t replica);   // owned by the pool
void* rlnamd_comm_handle(rlnamd_comm* c);   // the ncclComm_t
int rlnamd_comm_size(rlnamd_comm* c);
