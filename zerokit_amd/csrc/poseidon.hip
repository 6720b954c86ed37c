#include "poseidon.h"

#include <mutex>

#include <string.h>

namespace rlnamd {

// ---------------------------------------------------------------- Grain LFSR (poseidon_constants.rs:15-205)
namespace {
struct Grain {
  bool st[80];
  int head = 0;
  int nbits;
  Grain(int prime_bits, int t, int rf, int rp) : nbits(prime_bits) {
    memset(st, 0, sizeof(st));
    st[1] = true;  // field; s-box bits stay 0 (x^alpha)
    auto put = [&](int lo, int hi, uint64_t v) {
      for (int i = hi; i >= lo; i--) {
        st[i] = v & 1;
        v >>= 1;
      }
    };
    put(6, 17, prime_bits);
    put(18, 29, t);
    put(30, 39, rf);
    put(40, 49, rp);
    for (int i = 50; i < 80; i++) st[i] = true;
    for (int i = 0; i < 160; i++) update();
  }
  bool update() {
    bool b = st[(head + 62) % 80] ^ st[(head + 51) % 80] ^ st[(head + 38) % 80] ^ st[(head + 23) % 80] ^
             st[(head + 13) % 80] ^ st[head];
    st[head] = b;
    head = (head + 1) % 80;
    return b;
  }
  // n bits, first generated bit is the most significant (get_bits + reverse + LE packing in the reference)
  void value(uint32_t* limbs) {
    for (int i = 0; i < 8; i++) limbs[i] = 0;
    for (int k = nbits - 1; k >= 0; k--) {
      bool b = update();
      while (!b) {
        update();
        b = update();
      }
      if (update()) limbs[k >> 5] |= 1u << (k & 31);
    }
  }
  Fr rejection() {
    uint32_t v[8];
    for (;;) {
      value(v);
      if (!limbs_geq(v, FrParams::MOD)) return Fr::from_canonical(v);
    }
  }
  Fr mod_p() {
    uint32_t v[8];
    value(v);  // < 2^254 < 2r: one conditional subtraction == from_le_bytes_mod_order
    if (limbs_geq(v, FrParams::MOD)) Fr::reduce_once(v);
    return Fr::from_canonical(v);
  }
};
// rln/src/hashers.rs:14-23
const int kRoundParams[][4] = {{2, 8, 56, 0}, {3, 8, 57, 0}, {4, 8, 56, 0}, {5, 8, 60, 0},
                               {6, 8, 60, 0}, {7, 8, 63, 0}, {8, 8, 64, 0}, {9, 8, 63, 0}};
}  // namespace

// n x n inverse over Fr (Gauss-Jordan); the blocks Mhat_r of an MDS-derived matrix are invertible
static std::vector<Fr> mat_inv(std::vector<Fr> a, int n) {
  std::vector<Fr> b((size_t)n * n, Fr::zero());
  for (int i = 0; i < n; i++) b[i * n + i] = Fr::one();
  for (int c = 0; c < n; c++) {
    int p = c;
    while (p < n && a[p * n + c].is_zero()) p++;
    if (p == n) throw Error("Poseidon: singular block in the sparse round derivation");
    for (int j = 0; j < n; j++) {
      std::swap(a[c * n + j], a[p * n + j]);
      std::swap(b[c * n + j], b[p * n + j]);
    }
    const Fr iv = a[c * n + c].inv();
    for (int j = 0; j < n; j++) {
      a[c * n + j] = a[c * n + j] * iv;
      b[c * n + j] = b[c * n + j] * iv;
    }
    for (int r = 0; r < n; r++) {
      if (r == c || a[r * n + c].is_zero()) continue;
      const Fr f = a[r * n + c];
      for (int j = 0; j < n; j++) {
        a[r * n + j] = a[r * n + j] - f * a[c * n + j];
        b[r * n + j] = b[r * n + j] - f * b[c * n + j];
      }
    }
  }
  return b;
}

// the equivalent sparse form of the partial rounds (poseidon.h header comment)
static void derive_sparse_rounds(PoseidonParams& P) {
  const int t = P.t, half = P.rf / 2, n = t - 1;
  const std::vector<Fr>& M = P.mds;
  // constants pushed forward
  std::vector<Fr> v(t, Fr::zero());
  for (int r = 0; r < P.rp; r++) {
    const Fr* c = P.ark.data() + (size_t)(half + r) * t;
    std::vector<Fr> k(t);
    for (int j = 0; j < t; j++) k[j] = c[j] + v[j];
    P.k0.push_back(k[0]);
    for (int i = 0; i < t; i++) {
      Fr acc = Fr::zero();
      for (int j = 1; j < t; j++) acc = acc + M[i * t + j] * k[j];
      v[i] = acc;
    }
  }
  P.ark2.assign(P.ark.begin() + (size_t)(half + P.rp) * t, P.ark.end());
  for (int j = 0; j < t; j++) P.ark2[j] = P.ark2[j] + v[j];
  // matrices: M_r = M A_(r-1) = A_r B_r
  std::vector<Fr> A((size_t)t * t, Fr::zero());
  for (int i = 0; i < t; i++) A[i * t + i] = Fr::one();
  for (int r = 0; r < P.rp; r++) {
    std::vector<Fr> Mr((size_t)t * t, Fr::zero());
    for (int i = 0; i < t; i++)
      for (int j = 0; j < t; j++) {
        Fr acc = Fr::zero();
        for (int l = 0; l < t; l++) acc = acc + M[i * t + l] * A[l * t + j];
        Mr[i * t + j] = acc;
      }
    std::vector<Fr> hat((size_t)n * n), w(n);
    for (int i = 0; i < n; i++) {
      w[i] = Mr[(i + 1) * t];
      for (int j = 0; j < n; j++) hat[i * n + j] = Mr[(i + 1) * t + j + 1];
    }
    for (int j = 0; j < t; j++) P.row0.push_back(Mr[j]);
    const std::vector<Fr> hi = mat_inv(hat, n);
    for (int i = 0; i < n; i++) {
      Fr acc = Fr::zero();
      for (int j = 0; j < n; j++) acc = acc + hi[i * n + j] * w[j];
      P.u.push_back(acc);
    }
    std::fill(A.begin(), A.end(), Fr::zero());
    A[0] = Fr::one();
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) A[(i + 1) * t + j + 1] = hat[i * n + j];
    if (r == P.rp - 1) P.a_fin = hat;
  }
}

PoseidonParams poseidon_derive_params(int t) {
  const int* rp = nullptr;
  for (auto& p : kRoundParams)
    if (p[0] == t) rp = p;
  if (!rp) throw Error("no Poseidon parameters for input length " + std::to_string(t - 1));
  PoseidonParams P;
  P.t = t;
  P.rf = rp[1];
  P.rp = rp[2];
  Grain g(254, t, P.rf, P.rp);
  for (int i = 0; i < (P.rf + P.rp) * t; i++) P.ark.push_back(g.rejection());
  for (int k = 0; k < rp[3]; k++)
    for (int i = 0; i < 2 * t; i++) g.mod_p();
  std::vector<Fr> xs, ys;
  for (int i = 0; i < t; i++) xs.push_back(g.mod_p());
  for (int i = 0; i < t; i++) ys.push_back(g.mod_p());
  P.mds.resize(t * t);
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) P.mds[i * t + j] = (xs[i] + ys[j]).inv();
  derive_sparse_rounds(P);
  return P;
}

// Host evaluation of the parameter set in both forms (no device): the reference's dense rounds and the sparse
// partial rounds the kernels use.  Only the parameter self-check (rlnamd_poseidon_params_check) calls it.
void poseidon_params_eval_host(const PoseidonParams& P, const Fr* in, Fr* out_dense, Fr* out_sparse) {
  const int t = P.t, half = P.rf / 2;
  auto pow5 = [](const Fr& x) { Fr x2 = x.sqr(); return x2.sqr() * x; };
  auto mix = [&](std::vector<Fr>& s) {
    std::vector<Fr> n(t);
    for (int i = 0; i < t; i++) {
      Fr acc = Fr::zero();
      for (int j = 0; j < t; j++) acc = acc + P.mds[i * t + j] * s[j];
      n[i] = acc;
    }
    s = n;
  };
  std::vector<Fr> s(t, Fr::zero());
  for (int j = 1; j < t; j++) s[j] = in[j - 1];
  for (int r = 0; r < P.rf + P.rp; r++) {  // poseidon_hash.rs:117-133
    for (int j = 0; j < t; j++) s[j] = s[j] + P.ark[(size_t)r * t + j];
    const bool full = r < half || r >= half + P.rp;
    for (int j = 0; j < (full ? t : 1); j++) s[j] = pow5(s[j]);
    mix(s);
  }
  *out_dense = s[0];
  std::fill(s.begin(), s.end(), Fr::zero());
  for (int j = 1; j < t; j++) s[j] = in[j - 1];
  for (int r = 0; r < half; r++) {
    for (int j = 0; j < t; j++) s[j] = pow5(s[j] + P.ark[(size_t)r * t + j]);
    mix(s);
  }
  for (int r = 0; r < P.rp; r++) {
    const Fr x0 = pow5(s[0] + P.k0[r]);
    Fr n0 = P.row0[(size_t)r * t] * x0;
    for (int j = 1; j < t; j++) n0 = n0 + P.row0[(size_t)r * t + j] * s[j];
    for (int j = 1; j < t; j++) s[j] = s[j] + P.u[(size_t)r * (t - 1) + j - 1] * x0;
    s[0] = n0;
  }
  {
    std::vector<Fr> n(t);
    for (int i = 1; i < t; i++) {
      Fr acc = Fr::zero();
      for (int j = 1; j < t; j++) acc = acc + P.a_fin[(size_t)(i - 1) * (t - 1) + j - 1] * s[j];
      n[i] = acc;
    }
    for (int i = 1; i < t; i++) s[i] = n[i];
  }
  for (int r = 0; r < half; r++) {
    for (int j = 0; j < t; j++) s[j] = pow5(s[j] + P.ark2[(size_t)r * t + j]);
    mix(s);
  }
  *out_sparse = s[0];
}

Fr poseidon_hash_host(const PoseidonParams& P, const Fr* in) {
  const int t = P.t, half = P.rf / 2;
  if (t > POSEIDON_MAX_T) throw Error("unsupported Poseidon width");
  Fr s[POSEIDON_MAX_T], n[POSEIDON_MAX_T];
  auto pow5 = [](const Fr& x) { const Fr x2 = x.sqr(); return x2.sqr() * x; };
  auto mix = [&]() {
    for (int i = 0; i < t; i++) {
      Fr acc = P.mds[i * t] * s[0];
      for (int j = 1; j < t; j++) acc = acc + P.mds[i * t + j] * s[j];
      n[i] = acc;
    }
    for (int i = 0; i < t; i++) s[i] = n[i];
  };
  s[0] = Fr::zero();
  for (int j = 1; j < t; j++) s[j] = in[j - 1];
  for (int r = 0; r < half; r++) {
    for (int j = 0; j < t; j++) s[j] = pow5(s[j] + P.ark[(size_t)r * t + j]);
    mix();
  }
  for (int r = 0; r < P.rp; r++) {
    const Fr x0 = pow5(s[0] + P.k0[r]);
    const Fr* row = P.row0.data() + (size_t)r * t;
    const Fr* u = P.u.data() + (size_t)r * (t - 1);
    Fr n0 = row[0] * x0;
    for (int j = 1; j < t; j++) n0 = n0 + row[j] * s[j];
    for (int j = 1; j < t; j++) s[j] = s[j] + u[j - 1] * x0;
    s[0] = n0;
  }
  for (int i = 1; i < t; i++) {
    Fr acc = Fr::zero();
    for (int j = 1; j < t; j++) acc = acc + P.a_fin[(size_t)(i - 1) * (t - 1) + j - 1] * s[j];
    n[i] = acc;
  }
  for (int i = 1; i < t; i++) s[i] = n[i];
  for (int r = 0; r < half; r++) {
    for (int j = 0; j < t; j++) s[j] = pow5(s[j] + P.ark2[(size_t)r * t + j]);
    if (r + 1 < half) {
      mix();
    } else {   // only lane 0 of the last mix is the hash
      Fr acc = P.mds[0] * s[0];
      for (int j = 1; j < t; j++) acc = acc + P.mds[j] * s[j];
      s[0] = acc;
    }
  }
  return s[0];
}

const PoseidonParams& poseidon_host_params(int t) {
  if (t < 2 || t > POSEIDON_MAX_T) throw Error("unsupported Poseidon width t=" + std::to_string(t));
  static std::mutex mu;
  static PoseidonParams cache[POSEIDON_MAX_T + 1];
  std::lock_guard<std::mutex> lk(mu);
  if (cache[t].t == 0) cache[t] = poseidon_derive_params(t);
  return cache[t];
}

__global__ void k_fr_to29(const Fr* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fr29 v = Fr29::from_fq(src[t]);
  v.normalize();
#pragma unroll
  for (int k = 0; k < 9; k++) dst[t * 9 + k] = v.v[k];
}

void PoseidonDev::init() {
  if (ready) return;
  require_gpu();
  for (int t = 2; t <= POSEIDON_MAX_T; t++) {
    PoseidonParams P = poseidon_derive_params(t);
    rf[t] = P.rf;
    rp[t] = P.rp;
    ark[t].alloc(P.ark.size());
    mds[t].alloc(P.mds.size());
    RLN_HIP(hipMemcpy(ark[t].p, P.ark.data(), P.ark.size() * sizeof(Fr), hipMemcpyHostToDevice));
    RLN_HIP(hipMemcpy(mds[t].p, P.mds.data(), P.mds.size() * sizeof(Fr), hipMemcpyHostToDevice));
    ark29[t].alloc(P.ark.size() * 9);
    mds29[t].alloc(P.mds.size() * 9);
    hipLaunchKernelGGL(k_fr_to29, dim3(div_up(P.ark.size(), 64)), dim3(64), 0, 0, ark[t].p, ark29[t].p, (uint32_t)P.ark.size());
    hipLaunchKernelGGL(k_fr_to29, dim3(div_up(P.mds.size(), 64)), dim3(64), 0, 0, mds[t].p, mds29[t].p, (uint32_t)P.mds.size());
    // sparse partial rounds: k0 | row0 | u | a_fin | ark2
    std::vector<Fr> all;
    off_k0[t] = all.size(); all.insert(all.end(), P.k0.begin(), P.k0.end());
    off_row0[t] = all.size(); all.insert(all.end(), P.row0.begin(), P.row0.end());
    off_u[t] = all.size(); all.insert(all.end(), P.u.begin(), P.u.end());
    off_afin[t] = all.size(); all.insert(all.end(), P.a_fin.begin(), P.a_fin.end());
    off_ark2[t] = all.size(); all.insert(all.end(), P.ark2.begin(), P.ark2.end());
    DevBuf<Fr> tmp(all.size());
    RLN_HIP(hipMemcpy(tmp.p, all.data(), all.size() * sizeof(Fr), hipMemcpyHostToDevice));
    opt29[t].alloc(all.size() * 9);
    hipLaunchKernelGGL(k_fr_to29, dim3(div_up(all.size(), 64)), dim3(64), 0, 0, tmp.p, opt29[t].p, (uint32_t)all.size());
    RLN_HIP(hipDeviceSynchronize());
  }
  ready = true;
}

// one constant set per device: hipMalloc'ed memory belongs to the device that was current, and a process may drive
// several (rlnamd_pool: one replica per GPU)
PoseidonDev& poseidon_dev() {
  static std::mutex mu;
  static PoseidonDev per_device[64];
  int dev = 0;
  RLN_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) throw Error("device ordinal out of range");
  std::lock_guard<std::mutex> lk(mu);
  per_device[dev].init();
  return per_device[dev];
}

PoseidonView poseidon_view(int t) {
  if (t < 2 || t > POSEIDON_MAX_T) throw Error("unsupported Poseidon width t=" + std::to_string(t));
  PoseidonDev& d = poseidon_dev();
  const uint32_t* o = d.opt29[t].p;
  return {d.ark[t].p, d.mds[t].p, d.rf[t], d.rp[t], d.ark29[t].p, d.mds29[t].p, o + 9 * d.off_k0[t], o + 9 * d.off_row0[t],
          o + 9 * d.off_u[t], o + 9 * d.off_afin[t], o + 9 * d.off_ark2[t]};
}

template <int T>
__global__ void __launch_bounds__(256) k_poseidon_batch(const uint32_t* __restrict__ in, size_t n,
                                                        uint32_t* __restrict__ out, PoseidonView pv) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr x[T - 1];
#pragma unroll
  for (int j = 0; j < T - 1; j++) x[j] = Fr::from_canonical(in + (i * (T - 1) + j) * 8);
  Fr h = poseidon_hash_dev<T>(x, pv);
  h.to_canonical(out + i * 8);
}

void poseidon_hash_batch_device(const uint8_t* d_in, size_t n, int arity, uint8_t* d_out, hipStream_t s) {
  if (n == 0) return;
  PoseidonView pv = poseidon_view(arity + 1);
  dim3 grid(div_up(n, 256)), block(256);
  const uint32_t* in = (const uint32_t*)d_in;
  uint32_t* out = (uint32_t*)d_out;
  switch (arity) {
    case 1: hipLaunchKernelGGL(k_poseidon_batch<2>, grid, block, 0, s, in, n, out, pv); break;
    case 2: hipLaunchKernelGGL(k_poseidon_batch<3>, grid, block, 0, s, in, n, out, pv); break;
    case 3: hipLaunchKernelGGL(k_poseidon_batch<4>, grid, block, 0, s, in, n, out, pv); break;
    case 4: hipLaunchKernelGGL(k_poseidon_batch<5>, grid, block, 0, s, in, n, out, pv); break;
    case 5: hipLaunchKernelGGL(k_poseidon_batch<6>, grid, block, 0, s, in, n, out, pv); break;
    case 6: hipLaunchKernelGGL(k_poseidon_batch<7>, grid, block, 0, s, in, n, out, pv); break;
    case 7: hipLaunchKernelGGL(k_poseidon_batch<8>, grid, block, 0, s, in, n, out, pv); break;
    case 8: hipLaunchKernelGGL(k_poseidon_batch<9>, grid, block, 0, s, in, n, out, pv); break;
    default: throw Error("unsupported Poseidon arity " + std::to_string(arity));
  }
  RLN_HIP(hipGetLastError());
}

}  // namespace rlnamd
