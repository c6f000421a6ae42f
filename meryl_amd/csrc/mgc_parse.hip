// mgc_parse.hip -- FASTA / FASTQ text -> base stream, on the device (gfx950).
//
// Replaces, for text that is already in HBM, the byte-at-a-time host loader
//   dnaSeqFile::loadBases (absent submodule) behind merylInput::loadBases, src/meryl/merylInput.C:245-271
//   + the chunk assembly with '.' breakers of the sweatShop loader, src/meryl/merylOp-countThreads.C:138-231
// Output = what the session's base buffer holds after mgc_push_bases of every sequence: the bases of the
// sequence lines with '\n' '\r' ' ' '\t' removed and one '.' where a record starts (a breaker between
// sequences; any non-ACGTacgt byte resets the k-mer downstream, so headers and qualities must simply not
// appear).  FASTA: any line structure.  FASTQ: strict four-line records (the device cannot know where a
// multi-line record ends without the sequential state machine); every '@' / '+' line start is validated and
// a violation sets an error flag -- the caller then re-reads that file with the host parser (meryl_seq.cpp).
//
// A finite-state parse made parallel the usual way: tiles of 16 KiB; pass 1 summarises every tile as a
// function of the (two / four) states it can be entered in, a single workgroup composes the summaries into
// each tile's entry state and output offset, pass 2 re-parses with that knowledge and writes.  The running
// state (output length, line type, "previous byte was a newline", error) lives in device memory, so
// consecutive chunks of a file are parsed without any host round trip.
#include "mgc_device.h"

namespace mgc {

typedef unsigned char      u8;
typedef unsigned int       u32;
typedef unsigned long long u64;

constexpr int PT_BLOCK = 1024;
constexpr int PT_BYTES = 16;                      // per thread: one 16-byte load
constexpr int PT_TILE  = PT_BLOCK * PT_BYTES;     // 16 KiB

struct ParseState {             // device-resident, one per session
  u64 out_len;                  // bytes of base stream written so far
  u64 file_start_len;           // out_len when the current file began (mgc_text_rollback)
  u32 state;                    // FASTA: 0 = in a header (or before the first one), 1 = in sequence lines
                                // FASTQ: type (0..3) of the line the next byte belongs to
  u32 prev_nl;                  // the previous byte was '\n' (the next byte starts a line)
  u32 error;                    // FASTQ structure violated
  u32 pad;
};

struct TileSummary {            // 16 x u32
  u32 nl;                       // FASTQ: newlines in the tile
  u32 c[4];                     // FASTQ: emit-eligible bytes by (newlines before them) & 3
  u32 hs[4];                    // FASTQ: line starts by class
  u32 bad_at, bad_plus;         // FASTQ: bit j: a line start of class j is not '@' / '+'
  u32 has_ls, last_kind;        // FASTA: the tile holds a line start; the last one is a header
  u32 eA, eB;                   // FASTA: bytes emitted regardless of the entry state / only if entered in sequence state
  u32 last_nl;                  // the tile's last byte is '\n'
};

struct TileInfo { u64 out_off; u32 state_in; u32 pad; };

__device__ __forceinline__ u32 p_lane() { return threadIdx.x & 63u; }
__device__ __forceinline__ u32 p_wave() { return threadIdx.x >> 6; }

template <typename T>
__device__ __forceinline__ T pblock_excl_sum(T v, T *s_tmp, T *total) {
  constexpr int NW = PT_BLOCK / 64;
  T x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { T y = __shfl_up(x, d); if ((int)p_lane() >= d) x += y; }
  __syncthreads();
  if (p_lane() == 63) s_tmp[p_wave()] = x;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) { T t = s_tmp[i]; if (i < (int)p_wave()) base += t; tot += t; }
  *total = tot;
  return base + x - v;
}
// exclusive running maximum (0 = nothing yet)
__device__ __forceinline__ u32 pblock_excl_max(u32 v, u32 *s_tmp, u32 *total) {
  constexpr int NW = PT_BLOCK / 64;
  u32 x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { u32 y = __shfl_up(x, d); if ((int)p_lane() >= d) x = x > y ? x : y; }
  __syncthreads();
  if (p_lane() == 63) s_tmp[p_wave()] = x;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) { u32 t = s_tmp[i]; if (i < (int)p_wave()) base = base > t ? base : t; tot = tot > t ? tot : t; }
  *total = tot;
  u32 up = __shfl_up(x, 1);
  if (p_lane() == 0) up = 0;
  return base > up ? base : up;
}

// What every pass needs to know about this thread's 16 bytes.
struct ThreadText {
  u8  b[PT_BYTES];
  u32 valid, nl, ws, ls;        // bit j = byte j
};

__device__ __forceinline__ void load_thread_text(const u8 *__restrict__ text, u64 n, u64 tile_base, u32 carry_prev_nl,
                                                 u8 *s_last, ThreadText &t) {
  const u64 i0 = tile_base + (u64)threadIdx.x * PT_BYTES;
  t.valid = 0; t.nl = 0; t.ws = 0;
  if (i0 + PT_BYTES <= n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(text + i0);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < PT_BYTES; j++) t.b[j] = (u8)(w[j >> 2] >> (8 * (j & 3)));
    t.valid = 0xFFFFu;
  } else {
#pragma unroll
    for (int j = 0; j < PT_BYTES; j++) {
      const bool ok = i0 + j < n;
      t.b[j] = ok ? text[i0 + j] : (u8)0;
      t.valid |= (ok ? 1u : 0u) << j;
    }
  }
#pragma unroll
  for (int j = 0; j < PT_BYTES; j++) {
    const u8 c = t.b[j];
    t.nl |= (c == '\n' ? 1u : 0u) << j;
    t.ws |= ((c == '\n' || c == '\r' || c == ' ' || c == '\t') ? 1u : 0u) << j;
  }
  t.nl &= t.valid;
  s_last[threadIdx.x] = t.b[PT_BYTES - 1];
  __syncthreads();
  u32 prev_is_nl;
  if (threadIdx.x > 0)      prev_is_nl = (s_last[threadIdx.x - 1] == '\n') ? 1u : 0u;
  else if (tile_base > 0)   prev_is_nl = (text[tile_base - 1] == '\n') ? 1u : 0u;
  else                      prev_is_nl = carry_prev_nl;
  t.ls = ((t.nl << 1) | prev_is_nl) & t.valid;
}

// ---- pass 1 ------------------------------------------------------------------------------------
template <bool FASTQ>
__global__ __launch_bounds__(PT_BLOCK)
void text_summary_kernel(const u8 *__restrict__ text, u64 n, const ParseState *__restrict__ ps, TileSummary *__restrict__ sums) {
  __shared__ u8  s_last[PT_BLOCK];
  __shared__ u32 s_tmp[PT_BLOCK / 64 + 1];
  __shared__ u32 s_acc[16];
  const u64 tile = blockIdx.x, tile_base = tile * (u64)PT_TILE;
  if (threadIdx.x < 16) s_acc[threadIdx.x] = 0;
  ThreadText t;
  load_thread_text(text, n, tile_base, ps->prev_nl, s_last, t);      // contains a barrier: s_acc is cleared
  if (FASTQ) {
    u32 tot;
    const u32 nb = pblock_excl_sum<u32>(__popc(t.nl), s_tmp, &tot);
    u32 c[4] = {0, 0, 0, 0}, hs[4] = {0, 0, 0, 0}, bad_at = 0, bad_plus = 0;
#pragma unroll
    for (int j = 0; j < PT_BYTES; j++) {
      if (!((t.valid >> j) & 1u)) continue;
      const u32 cls = (nb + __popc(t.nl & ((1u << j) - 1u))) & 3u;
      if (!((t.ws >> j) & 1u)) c[cls]++;
      if ((t.ls >> j) & 1u) {
        hs[cls]++;
        if (t.b[j] != '@') bad_at |= 1u << cls;
        if (t.b[j] != '+') bad_plus |= 1u << cls;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (c[q])  atomicAdd(&s_acc[q], c[q]);
      if (hs[q]) atomicAdd(&s_acc[4 + q], hs[q]);
    }
    if (bad_at)   atomicOr(&s_acc[8], bad_at);
    if (bad_plus) atomicOr(&s_acc[9], bad_plus);
    __syncthreads();
    if (threadIdx.x == 0) {
      TileSummary s = {};
      s.nl = tot;
      for (int q = 0; q < 4; q++) { s.c[q] = s_acc[q]; s.hs[q] = s_acc[4 + q]; }
      s.bad_at = s_acc[8]; s.bad_plus = s_acc[9];
      const u64 last = (tile_base + PT_TILE <= n) ? tile_base + PT_TILE - 1 : n - 1;
      s.last_nl = (text[last] == '\n') ? 1u : 0u;
      sums[tile] = s;
    }
  } else {
    // marker of the thread's last line start: position-ordered, kind in the low bit
    u32 mark = 0;
    if (t.ls) {
      const int j = 31 - __clz(t.ls);
      mark = (threadIdx.x + 1) * 2 + (t.b[j] == '>' ? 1u : 0u);
    }
    u32 last_mark;
    const u32 before = pblock_excl_max(mark, s_tmp, &last_mark);
    // bytes before the tile's first line start inherit the entry state
    u32 eA = 0, eB = 0;
    bool known = before != 0;
    u32  st = before & 1u ? 0u : 1u;                       // header -> 0, sequence -> 1 (only if known)
#pragma unroll
    for (int j = 0; j < PT_BYTES; j++) {
      if (!((t.valid >> j) & 1u)) continue;
      const bool ls = (t.ls >> j) & 1u, hd = ls && t.b[j] == '>';
      if (ls) { known = true; st = hd ? 0u : 1u; }
      if (hd) eA++;                                        // the '.' of a record start
      else if (!((t.ws >> j) & 1u)) { if (!known) eB++; else if (st) eA++; }
    }
    if (eA) atomicAdd(&s_acc[0], eA);
    if (eB) atomicAdd(&s_acc[1], eB);
    __syncthreads();
    if (threadIdx.x == 0) {
      TileSummary s = {};
      s.has_ls = last_mark != 0; s.last_kind = last_mark & 1u;
      s.eA = s_acc[0]; s.eB = s_acc[1];
      const u64 last = (tile_base + PT_TILE <= n) ? tile_base + PT_TILE - 1 : n - 1;
      s.last_nl = (text[last] == '\n') ? 1u : 0u;
      sums[tile] = s;
    }
  }
}

// ---- compose: one workgroup walks the tiles in batches of PT_BLOCK -----------------------------------
template <bool FASTQ>
__global__ __launch_bounds__(PT_BLOCK)
void text_compose_kernel(const TileSummary *__restrict__ sums, u64 ntiles, ParseState *__restrict__ ps, TileInfo *__restrict__ info) {
  __shared__ u32 s_tmp[PT_BLOCK / 64 + 1];
  __shared__ u64 s_tmp64[PT_BLOCK / 64 + 1];
  u64 out = 0;                                             // relative to ps->out_len
  u32 state = ps->state, err = 0, last_nl = ps->prev_nl;
  for (u64 base = 0; base < ntiles; base += PT_BLOCK) {
    const u64 t = base + threadIdx.x;
    const bool live = t < ntiles;
    TileSummary s = {};
    if (live) s = sums[t];
    u32 st_in;
    u64 emit;
    if (FASTQ) {
      u32 tot;
      const u32 nb = pblock_excl_sum<u32>(live ? s.nl : 0u, s_tmp, &tot);
      st_in = (state + nb) & 3u;
      emit = live ? (u64)s.c[(1u - st_in) & 3u] + s.hs[(0u - st_in) & 3u] : 0ull;
      if (live && (((s.bad_at >> ((0u - st_in) & 3u)) & 1u) || ((s.bad_plus >> ((2u - st_in) & 3u)) & 1u))) err = 1;
      state = (state + tot) & 3u;
    } else {
      const u32 mark = (live && s.has_ls) ? (threadIdx.x + 1) * 2 + s.last_kind : 0u;
      u32 last_mark;
      const u32 before = pblock_excl_max(mark, s_tmp, &last_mark);
      st_in = before ? ((before & 1u) ? 0u : 1u) : state;
      emit = live ? (u64)s.eA + (st_in ? s.eB : 0u) : 0ull;
      if (last_mark) state = (last_mark & 1u) ? 0u : 1u;
    }
    u64 tot64;
    const u64 off = pblock_excl_sum<u64>(emit, s_tmp64, &tot64);
    if (live) { TileInfo ti; ti.out_off = out + off; ti.state_in = st_in; ti.pad = 0; info[t] = ti; }
    out += tot64;
    if (base + PT_BLOCK >= ntiles) {                       // the last tile's trailing byte
      const u64 lt = ntiles - 1 - base;
      __syncthreads();
      if (threadIdx.x == lt) s_tmp[0] = s.last_nl;
      __syncthreads();
      last_nl = s_tmp[0];
    }
  }
  // every thread saw the same totals except `err`
  __syncthreads();
  if (threadIdx.x == 0) s_tmp[0] = 0;
  __syncthreads();
  if (err) atomicOr(&s_tmp[0], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    // pass 2 reads the OLD out_len as its base: publish the new one through `info[ntiles]`
    TileInfo fin; fin.out_off = out; fin.state_in = state; fin.pad = last_nl | (s_tmp[0] << 1);
    info[ntiles] = fin;
  }
}

// ---- pass 2 ------------------------------------------------------------------------------------
template <bool FASTQ>
__global__ __launch_bounds__(PT_BLOCK)
void text_emit_kernel(const u8 *__restrict__ text, u64 n, const ParseState *__restrict__ ps, const TileInfo *__restrict__ info,
                      u8 *__restrict__ out) {
  __shared__ u8  s_last[PT_BLOCK];
  __shared__ u32 s_tmp[PT_BLOCK / 64 + 1];
  const u64 tile = blockIdx.x, tile_base = tile * (u64)PT_TILE;
  const TileInfo ti = info[tile];
  ThreadText t;
  load_thread_text(text, n, tile_base, ps->prev_nl, s_last, t);
  u32 emit = 0, dot = 0;                                   // bit j: byte j is emitted / emitted as '.'
  if (FASTQ) {
    u32 tot;
    const u32 nb = pblock_excl_sum<u32>(__popc(t.nl), s_tmp, &tot);
#pragma unroll
    for (int j = 0; j < PT_BYTES; j++) {
      if (!((t.valid >> j) & 1u)) continue;
      const u32 type = (ti.state_in + nb + __popc(t.nl & ((1u << j) - 1u))) & 3u;
      if (((t.ls >> j) & 1u) && type == 0) { emit |= 1u << j; dot |= 1u << j; }
      else if (type == 1 && !((t.ws >> j) & 1u)) emit |= 1u << j;
    }
  } else {
    u32 mark = 0;
    if (t.ls) {
      const int j = 31 - __clz(t.ls);
      mark = (threadIdx.x + 1) * 2 + (t.b[j] == '>' ? 1u : 0u);
    }
    u32 last_mark;
    const u32 before = pblock_excl_max(mark, s_tmp, &last_mark);
    u32 st = before ? ((before & 1u) ? 0u : 1u) : ti.state_in;
#pragma unroll
    for (int j = 0; j < PT_BYTES; j++) {
      if (!((t.valid >> j) & 1u)) continue;
      const bool ls = (t.ls >> j) & 1u, hd = ls && t.b[j] == '>';
      if (ls) st = hd ? 0u : 1u;
      if (hd) { emit |= 1u << j; dot |= 1u << j; }
      else if (st && !((t.ws >> j) & 1u)) emit |= 1u << j;
    }
  }
  u32 tot;
  const u32 off = pblock_excl_sum<u32>(__popc(emit), s_tmp, &tot);
  u8 *o = out + ps->out_len + ti.out_off + off;
#pragma unroll
  for (int j = 0; j < PT_BYTES; j++)
    if ((emit >> j) & 1u) *o++ = ((dot >> j) & 1u) ? (u8)'.' : t.b[j];
}

// after pass 2 of a chunk: fold the chunk's totals into the running state
__global__ void text_commit_kernel(ParseState *ps, const TileInfo *info, u64 ntiles) {
  const TileInfo fin = info[ntiles];
  ps->out_len += fin.out_off;
  ps->state    = fin.state_in;
  ps->prev_nl  = fin.pad & 1u;
  ps->error   |= (fin.pad >> 1) & 1u;
}

// file boundaries: begin (format-specific entry state), end (breaker), rollback (drop the file's output)
__global__ void text_file_kernel(ParseState *ps, u8 *out, int what) {
  if (what == 0) {          // begin
    ps->file_start_len = ps->out_len; ps->state = 0; ps->prev_nl = 1; ps->error = 0;
  } else if (what == 1) {   // end: one breaker after the last sequence
    out[ps->out_len] = (u8)'.'; ps->out_len += 1;
  } else if (what == 2) {   // rollback
    ps->out_len = ps->file_start_len; ps->error = 0;
  } else {                  // reset everything
    ps->out_len = 0; ps->file_start_len = 0; ps->state = 0; ps->prev_nl = 1; ps->error = 0;
  }
}

__global__ void text_set_len_kernel(ParseState *ps, u64 new_len, int rebase_file) {
  ps->out_len = new_len;
  if (rebase_file) ps->file_start_len = 0;
  else if (ps->file_start_len > new_len) ps->file_start_len = new_len;
}

hipError_t launch_text_set_len(void *d_state, uint64_t new_len, int rebase_file, hipStream_t st) {
  hipLaunchKernelGGL(text_set_len_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<ParseState *>(d_state), (u64)new_len, rebase_file);
  return hipGetLastError();
}

size_t text_parse_state_bytes() { return sizeof(ParseState); }
size_t text_parse_workspace_bytes(uint64_t n) {
  const uint64_t ntiles = (n + PT_TILE - 1) / PT_TILE;
  return (size_t)(ntiles + 1) * (sizeof(TileSummary) + sizeof(TileInfo)) + 256;
}

hipError_t launch_text_parse(const uint8_t *d_text, uint64_t n, int fastq, void *d_state, void *d_ws, uint8_t *d_out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const uint64_t ntiles = (n + PT_TILE - 1) / PT_TILE;
  ParseState  *ps   = reinterpret_cast<ParseState *>(d_state);
  TileSummary *sums = reinterpret_cast<TileSummary *>(d_ws);
  TileInfo    *info = reinterpret_cast<TileInfo *>(reinterpret_cast<unsigned char *>(d_ws) +
                                                   (((ntiles + 1) * sizeof(TileSummary) + 255) / 256) * 256);
  if (fastq) {
    hipLaunchKernelGGL(text_summary_kernel<true>, dim3((uint32_t)ntiles), dim3(PT_BLOCK), 0, st, d_text, (u64)n, ps, sums);
    hipLaunchKernelGGL(text_compose_kernel<true>, dim3(1), dim3(PT_BLOCK), 0, st, sums, (u64)ntiles, ps, info);
    hipLaunchKernelGGL(text_emit_kernel<true>, dim3((uint32_t)ntiles), dim3(PT_BLOCK), 0, st, d_text, (u64)n, ps, info, d_out);
  } else {
    hipLaunchKernelGGL(text_summary_kernel<false>, dim3((uint32_t)ntiles), dim3(PT_BLOCK), 0, st, d_text, (u64)n, ps, sums);
    hipLaunchKernelGGL(text_compose_kernel<false>, dim3(1), dim3(PT_BLOCK), 0, st, sums, (u64)ntiles, ps, info);
    hipLaunchKernelGGL(text_emit_kernel<false>, dim3((uint32_t)ntiles), dim3(PT_BLOCK), 0, st, d_text, (u64)n, ps, info, d_out);
  }
  hipLaunchKernelGGL(text_commit_kernel, dim3(1), dim3(1), 0, st, ps, info, (u64)ntiles);
  return hipGetLastError();
}

hipError_t launch_text_file_op(void *d_state, uint8_t *d_out, int what, hipStream_t st) {
  hipLaunchKernelGGL(text_file_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<ParseState *>(d_state), d_out, what);
  return hipGetLastError();
}

}  // namespace mgc
