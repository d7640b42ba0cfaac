// snapshot.hip — state snapshots (checkpoint / resume) for whole-grid AND row-slab handles, and euler_render on
// a row-slab handle.
//
// A snapshot is everything the reference keeps in file-scope variables (main.c:64-100, 204, 577): the four velocity fields, the
// five cell grids, g_precon, the marker array in order, the RNG state and the source latch - plus this build's frame counters.
// Little-endian; layouts documented in include/euler.h.
//   * a whole-grid handle writes ONE file (version 1, version 2 with the dye);
//   * a job of row slabs writes one PART file per rank (version 3: the rank's own rows, its markers WITH their keys = their
//     positions in the reference's array) and, rank 0, a small manifest under the given name listing the ranks' band ranges.
// Loading is independent of how the state was written: any handle - whole grid, or a slab of ANY partition - takes its rows
// (ghost rows included: the files of a job hold every row) and the markers inside its rows from whichever files hold them.  So a
// 3-rank job can be resumed on 2 ranks, or on one GPU, or with another (fluid-balanced) partition: the re-partitioning of a run
// whose water has moved is save + load.
#include "euler_dev.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

int eu_set_source_count(euler_sim* S, size_t nsrc);   // driver.hip

static uint64_t snap_fnv(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
struct SnapHeader {
  char magic[8]; uint32_t version; int32_t X, Y; uint32_t reserved;
  uint64_t n_markers, rng_state; int32_t source_exhausted; int32_t reserved2;
  uint64_t frames, total_substeps, total_pcg_iterations;
};
struct SlabHeader {      // follows SnapHeader in a version 3 part file
  int32_t nranks, rank, band_lo, band_hi, row_lo, row_hi;
  uint64_t n_loc;
};
struct ManifestHeader { char magic[8]; uint32_t version; int32_t X, Y, nranks; };   // then nranks x {int32 band_lo, band_hi}
static const int SNAP_F32[] = {EULER_F_U, EULER_F_V, EULER_F_UTMP, EULER_F_VTMP};
static const int SNAP_U8[] = {EULER_F_SOLID, EULER_F_SOURCE, EULER_F_SINK, EULER_F_COUNT, EULER_F_PREV_COUNT};
static const int SNAP_DYE[] = {EULER_F_DYE_R, EULER_F_DYE_G, EULER_F_DYE_B, EULER_F_DYE_RTMP, EULER_F_DYE_GTMP, EULER_F_DYE_BTMP};   // version 2 only

static std::string part_name(const char* path, int rank, int nranks) {
  char suffix[64];
  snprintf(suffix, sizeof suffix, ".%dof%d", rank, nranks);
  return std::string(path) + suffix;
}

// ------------------------------------------------------------------------------------------ save
// Files appear under their final names only when complete: every file is written under "<name>.tmp" and renamed.  Row slabs (collective): the ranks
// agree (one all-reduce of the status) that every part file was written and closed before any of them is renamed, and again that every rename
// succeeded before rank 0 puts the manifest in place - a manifest never names a part that is missing or belongs to an older state, and a rank
// that fails makes the call fail on every rank.
static int write_part_file(euler_sim* S, const euler_stats& st, const std::string& tmp, bool slab, int nranks, int rank) {
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) { eu_set_error("cannot open %s for writing", tmp.c_str()); return EULER_EIO; }
  int rc = EULER_OK;
  SnapHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "EULERSNP", 8);
  h.version = slab ? 3 : (S->dye[0] ? 2 : 1); h.reserved = (slab && S->dye[0]) ? 1 : 0;      // (version 3 with the dye: reserved = 1)
  h.X = S->X; h.Y = S->Y; h.n_markers = st.n_markers; h.rng_state = st.rng_state;
  h.source_exhausted = st.source_exhausted; h.frames = st.frames; h.total_substeps = st.total_substeps;
  h.total_pcg_iterations = st.total_pcg_iterations;
  uint64_t sum = snap_fnv(14695981039346656037ull, &h, sizeof h);
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  const uint64_t n_loc = S->n_markers_host;      // exact: euler_get_stats synced the marker state
  if (slab) {
    SlabHeader sh;
    memset(&sh, 0, sizeof sh);
    sh.nranks = nranks; sh.rank = rank; sh.band_lo = S->band_lo; sh.band_hi = S->band_hi; sh.row_lo = S->row_lo; sh.row_hi = S->row_hi; sh.n_loc = n_loc;
    sum = snap_fnv(sum, &sh, sizeof sh);
    ok = ok && fwrite(&sh, sizeof sh, 1, f) == 1;
  }
  const size_t Cr = (size_t)(S->row_hi - S->row_lo) * S->X;      // cells of the own rows (the whole grid without slabs)
  std::vector<unsigned char> buf;
  try { buf.resize(Cr * 8 > n_loc * 8 ? Cr * 8 : (size_t)n_loc * 8); } catch (...) { fclose(f); return EULER_ENOMEM; }
  auto put = [&](int field, size_t bytes) {
    if (!ok || !bytes) return;
    rc = euler_get_field(S, field, buf.data(), bytes);
    if (rc) { ok = false; return; }
    sum = snap_fnv(sum, buf.data(), bytes);
    ok = fwrite(buf.data(), 1, bytes, f) == bytes;
  };
  for (int fd : SNAP_F32) put(fd, Cr * 4);
  for (int fd : SNAP_U8) put(fd, Cr);
  put(EULER_F_PRECON, Cr * 8);
  if (S->dye[0]) for (int fd : SNAP_DYE) put(fd, Cr * 4);
  put(EULER_F_MARKERS, (size_t)n_loc * 8);
  if (slab) put(EULER_F_MARKER_KEYS, (size_t)n_loc * 4);
  ok = ok && fwrite(&sum, 8, 1, f) == 1;
  ok = (fclose(f) == 0) && ok;
  if (rc) return rc;
  if (!ok) { eu_set_error("short write to %s", tmp.c_str()); return EULER_EIO; }
  return EULER_OK;
}

extern "C" int euler_save_state(euler_sim* S, const char* path) {
  if (!S || !path) return EULER_EINVAL;
  euler_stats st;
  int rc = euler_get_stats(S, &st);
  const bool slab = S->slab_on != 0;
  if (rc && !slab) return rc;
  const int nranks = slab ? S->cfg.slab_nranks : 1, rank = slab ? S->cfg.slab_rank : 0;
  const std::string file = slab ? part_name(path, rank, nranks) : std::string(path);
  const std::string tmp = file + ".tmp";
  if (!rc) rc = write_part_file(S, st, tmp, slab, nranks, rank);
  auto agree = [&](int local, const char* what) -> int {      // row slabs: non-zero on every rank if any rank failed
    if (!slab || !S->has_comm) return local;
    int worst = local;
    const int rc2 = eu_slab_status_sync(S, local, &worst);
    if (rc2) return rc2;
    if (worst && !local) eu_set_error("euler_save_state(%s): another rank failed to %s", path, what);
    return worst;
  };
  if ((rc = agree(rc, "write its part file"))) { remove(tmp.c_str()); return rc; }
  if (rename(tmp.c_str(), file.c_str()) != 0) { eu_set_error("cannot rename %s to %s", tmp.c_str(), file.c_str()); rc = EULER_EIO; }
  if ((rc = agree(rc, "put its part file in place"))) return rc;
  if (slab && rank == 0) {      // the manifest: which part holds which bands - last, and complete when it appears
    const std::string mtmp = std::string(path) + ".tmp";
    FILE* m = fopen(mtmp.c_str(), "wb");
    bool mok = m != nullptr;
    if (mok) {
      ManifestHeader mh;
      memset(&mh, 0, sizeof mh);
      memcpy(mh.magic, "EULERMAN", 8);
      mh.version = 3; mh.X = S->X; mh.Y = S->Y; mh.nranks = nranks;
      mok = fwrite(&mh, sizeof mh, 1, m) == 1;
      for (int r = 0; r < nranks && mok; ++r) { const int32_t b[2] = {S->part_lo[r], S->part_hi[r]}; mok = fwrite(b, sizeof b, 1, m) == 1; }
      mok = (fclose(m) == 0) && mok;
      mok = mok && rename(mtmp.c_str(), path) == 0;
    }
    if (!mok) { eu_set_error("cannot write the manifest %s", path); rc = EULER_EIO; }
  }
  return agree(rc, "write the manifest");
}

// ------------------------------------------------------------------------------------------ load
namespace {
struct Part {
  std::string file;
  SnapHeader h;
  int row_lo, row_hi;               // rows the FILE holds
  int keep_lo, keep_hi;             // rows of it kept in memory: those the handle's window needs
  uint64_t n_loc;                   // markers in the file
  bool keyed;                       // version 3: markers carry keys; else a marker's key is its index
  // kept in memory: the rows [keep_lo, keep_hi) of every field section, and the markers whose row lies in [mark_lo, mark_hi) with their keys
  std::vector<unsigned char> sec[16];
  std::vector<float> mk;
  std::vector<unsigned int> keys;
  bool has_dye() const { return h.version == 2 || (h.version == 3 && h.reserved == 1); }
  const unsigned char* rows(int k, int y) const { return sec[k].data() + (size_t)(y - keep_lo) * h.X * elem(k); }      // row y (kept) of section k
  static size_t elem(int k) { return k <= 3 ? 4 : (k <= 8 ? 1 : (k == 9 ? 8 : 4)); }      // 0..3 f32 fields, 4..8 u8 grids, 9 precon, 10..15 dye
};

// One pass over the file in 4 MB pieces: the checksum covers every byte, memory holds only what the handle needs - the rows of its window and the markers
// inside [mark_lo, mark_hi) - so a whole-grid snapshot of 16384^2 (7.8 GB + 2.7 GB of markers) loaded into eight row slabs costs each rank its own eighth,
// not the whole file.
int read_part(Part& p, int need_lo, int need_hi, int mark_lo, int mark_hi) {
  FILE* f = fopen(p.file.c_str(), "rb");
  if (!f) { eu_set_error("cannot open %s", p.file.c_str()); return EULER_EIO; }
  struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < (long)(sizeof(SnapHeader) + 8) || fread(&p.h, sizeof p.h, 1, f) != 1 || memcmp(p.h.magic, "EULERSNP", 8) != 0 || p.h.version < 1 || p.h.version > 3) {
    eu_set_error("%s is not an euler state snapshot (version 1, 2 or 3)", p.file.c_str()); return EULER_EINVAL;
  }
  uint64_t sum = snap_fnv(14695981039346656037ull, &p.h, sizeof p.h);
  size_t body = sizeof(SnapHeader);
  p.row_lo = 0; p.row_hi = p.h.Y; p.n_loc = p.h.n_markers; p.keyed = false;
  if (p.h.version == 3) {
    SlabHeader sh;
    if ((size_t)n < sizeof(SnapHeader) + sizeof sh + 8 || fread(&sh, sizeof sh, 1, f) != 1) { eu_set_error("%s is truncated (slab header)", p.file.c_str()); return EULER_EIO; }
    sum = snap_fnv(sum, &sh, sizeof sh);
    body += sizeof sh;
    p.row_lo = sh.row_lo; p.row_hi = sh.row_hi; p.n_loc = sh.n_loc; p.keyed = true;
  }
  if (p.h.X <= 0 || p.row_hi < p.row_lo) { eu_set_error("%s: bad header", p.file.c_str()); return EULER_EINVAL; }
  const size_t X = (size_t)p.h.X, C = (size_t)(p.row_hi - p.row_lo) * X;
  const size_t want = body + C * 29 + (p.has_dye() ? C * 24 : 0) + (size_t)p.n_loc * (p.keyed ? 12 : 8) + 8;
  if ((size_t)n != want) { eu_set_error("%s is truncated or corrupt (size)", p.file.c_str()); return EULER_EIO; }
  p.keep_lo = need_lo > p.row_lo ? need_lo : p.row_lo; p.keep_hi = need_hi < p.row_hi ? need_hi : p.row_hi;
  if (p.keep_hi < p.keep_lo) p.keep_hi = p.keep_lo;
  std::vector<unsigned char> chunk;
  try { chunk.resize((size_t)4 << 20); } catch (...) { return EULER_ENOMEM; }
  // a stretch of `bytes` bytes of the file; `sink(offset in the stretch, data, length)` sees every piece in order
  auto stream = [&](size_t bytes, auto&& sink) -> bool {
    for (size_t done = 0; done < bytes;) {
      const size_t len = bytes - done < chunk.size() ? bytes - done : chunk.size();
      if (fread(chunk.data(), 1, len, f) != len) return false;
      sum = snap_fnv(sum, chunk.data(), len);
      sink(done, chunk.data(), len);
      done += len;
    }
    return true;
  };
  const int nsec = p.has_dye() ? 16 : 10;
  for (int k = 0; k < nsec; ++k) {
    const size_t e = Part::elem(k), lo = (size_t)(p.keep_lo - p.row_lo) * X * e, hi = (size_t)(p.keep_hi - p.row_lo) * X * e;      // kept bytes of the section
    try { p.sec[k].resize(hi - lo); } catch (...) { return EULER_ENOMEM; }
    unsigned char* dst = p.sec[k].data();
    const bool ok = stream(C * e, [&](size_t off, const unsigned char* d, size_t len) {
      const size_t a = off > lo ? off : lo, b2 = off + len < hi ? off + len : hi;
      if (b2 > a) memcpy(dst + (a - lo), d + (a - off), b2 - a);
    });
    if (!ok) { eu_set_error("%s: short read", p.file.c_str()); return EULER_EIO; }
  }
  // markers: those whose row lies in [mark_lo, mark_hi); the indices kept decide which keys are kept from the stretch behind them
  std::vector<uint64_t> kept;
  unsigned char carry[8];
  size_t ncarry = 0;
  uint64_t mi = 0;
  bool oom = false;
  bool ok = stream((size_t)p.n_loc * 8, [&](size_t, const unsigned char* d, size_t len) {
    size_t pos = 0;
    while (pos < len && !oom) {
      float m[2];
      if (ncarry || len - pos < 8) {      // a marker split across two pieces
        const size_t take = 8 - ncarry < len - pos ? 8 - ncarry : len - pos;
        memcpy(carry + ncarry, d + pos, take); ncarry += take; pos += take;
        if (ncarry < 8) break;
        memcpy(m, carry, 8); ncarry = 0;
      } else { memcpy(m, d + pos, 8); pos += 8; }
      const int y = (int)floorf(m[1] / EU_H);
      if (y >= mark_lo && y < mark_hi) {
        try { p.mk.push_back(m[0]); p.mk.push_back(m[1]); kept.push_back(mi); } catch (...) { oom = true; }
      }
      ++mi;
    }
  });
  if (oom) return EULER_ENOMEM;
  if (!ok) { eu_set_error("%s: short read", p.file.c_str()); return EULER_EIO; }
  if (p.keyed) {
    size_t next = 0;
    try { p.keys.resize(kept.size()); } catch (...) { return EULER_ENOMEM; }
    ncarry = 0; mi = 0;
    ok = stream((size_t)p.n_loc * 4, [&](size_t, const unsigned char* d, size_t len) {      // (4 MB pieces are whole multiples of 4 bytes)
      for (size_t pos = 0; pos + 4 <= len; pos += 4, ++mi)
        if (next < kept.size() && kept[next] == mi) { memcpy(&p.keys[next], d + pos, 4); ++next; }
    });
    if (!ok) { eu_set_error("%s: short read", p.file.c_str()); return EULER_EIO; }
  } else {
    try { p.keys.resize(kept.size()); } catch (...) { return EULER_ENOMEM; }
    for (size_t i = 0; i < kept.size(); ++i) p.keys[i] = (unsigned int)kept[i];
  }
  uint64_t stored = 0;
  if (fread(&stored, 8, 1, f) != 1 || stored != sum) { eu_set_error("%s is truncated or corrupt (checksum)", p.file.c_str()); return EULER_EIO; }
  return EULER_OK;
}
}  // namespace

// everything of a load that is this rank's own business (files, host buffers, uploads): NO collective in here, so a failure leaves no other rank waiting
static int load_state_local(euler_sim* S, const char* path, SnapHeader* h_out) {
  // what is there: one whole-grid file, or a manifest naming the part files of a job
  std::vector<Part> parts;
  {
    FILE* f = fopen(path, "rb");
    if (!f) { eu_set_error("cannot open %s", path); return EULER_EIO; }
    char magic[8] = {0};
    const bool got = fread(magic, 8, 1, f) == 1;
    if (got && memcmp(magic, "EULERMAN", 8) == 0) {
      ManifestHeader mh;
      fseek(f, 0, SEEK_SET);
      if (fread(&mh, sizeof mh, 1, f) != 1 || mh.version != 3 || mh.nranks < 1 || mh.nranks > 64) { fclose(f); eu_set_error("%s: bad manifest", path); return EULER_EINVAL; }
      for (int r = 0; r < mh.nranks; ++r) {
        int32_t b[2];
        if (fread(b, sizeof b, 1, f) != 1) { fclose(f); eu_set_error("%s: bad manifest", path); return EULER_EINVAL; }
        Part p;
        p.file = part_name(path, r, mh.nranks);
        p.row_lo = 64 * b[0]; p.row_hi = 64 * b[1] < mh.Y ? 64 * b[1] : mh.Y;      // (confirmed by the part's own header when it is read)
        parts.push_back(p);
      }
    } else {
      Part p;
      p.file = path; p.row_lo = 0; p.row_hi = S->Y;
      parts.push_back(p);
    }
    fclose(f);
  }
  // a handle needs the rows of its window (ghost rows included) and the markers inside its own rows: only the parts that overlap
  const int X = S->X, Y = S->Y;
  const int need_lo = S->slab_on ? S->win_lo : 0, need_hi = S->slab_on ? S->win_hi : Y;
  std::vector<Part*> use;
  for (Part& p : parts) {
    if (p.row_hi <= need_lo || p.row_lo >= need_hi) continue;
    int rc = read_part(p, need_lo, need_hi, S->slab_on ? S->row_lo : 0, S->slab_on ? S->row_hi : Y);
    if (rc) return rc;
    if (p.h.X != X || p.h.Y != Y) { eu_set_error("snapshot grid %dx%d does not fit this %dx%d handle", p.h.X, p.h.Y, X, Y); return EULER_EINVAL; }
    use.push_back(&p);
  }
  if (use.empty()) { eu_set_error("%s holds no rows of this handle", path); return EULER_EINVAL; }
  const SnapHeader& h0 = use[0]->h;
  const bool dye = use[0]->has_dye();
  if (dye != (S->dye[0] != nullptr)) {
    eu_set_error("%s %s the dye fields but this handle was created %s euler_config.rainbow", path, dye ? "carries" : "lacks", S->dye[0] ? "with" : "without");
    return EULER_EINVAL;
  }
  for (Part* p : use)
    if (p->h.n_markers != h0.n_markers || p->h.rng_state != h0.rng_state || p->h.frames != h0.frames) { eu_set_error("%s: the part files belong to different states", path); return EULER_EINVAL; }

  // ---- rows [need_lo, need_hi) of a field section from the parts
  const int rows = need_hi - need_lo;
  std::vector<unsigned char> buf((size_t)rows * X * 8);
  std::vector<char> have((size_t)rows);
  auto gather = [&](int section, int elem, int y0, int y1) -> bool {      // rows [y0, y1) -> buf
    std::fill(have.begin(), have.end(), 0);
    for (Part* p : use) {
      const int a = p->row_lo > y0 ? p->row_lo : y0, b = p->row_hi < y1 ? p->row_hi : y1;
      if (b <= a) continue;
      memcpy(buf.data() + (size_t)(a - y0) * X * elem, p->rows(section, a), (size_t)(b - a) * X * elem);
      for (int y = a; y < b; ++y) have[(size_t)(y - y0)] = 1;
    }
    for (int y = y0; y < y1; ++y) if (!have[(size_t)(y - y0)]) { eu_set_error("%s: row %d is in none of the part files", path, y); return false; }
    return true;
  };
  HIPCHK(hipStreamSynchronize(S->stream));
  float* f32[4] = {S->u, S->v, S->utmp, S->vtmp};
  uint8_t* u8[5] = {S->solid, S->source, S->sink, S->count, S->prev_count};
  for (int k = 0; k < 4; ++k) {
    if (!gather(k, 4, need_lo, need_hi)) return EULER_EINVAL;
    HIPCHK(hipMemcpy(f32[k] + (size_t)need_lo * X, buf.data(), (size_t)rows * X * 4, hipMemcpyHostToDevice));
  }
  size_t nsrc = 0;
  for (int k = 0; k < 5; ++k) {
    if (!gather(4 + k, 1, need_lo, need_hi)) return EULER_EINVAL;
    HIPCHK(hipMemcpy(u8[k] + (size_t)need_lo * X, buf.data(), (size_t)rows * X, hipMemcpyHostToDevice));
    if (k == 1) for (size_t i = (size_t)(S->row_lo - need_lo) * X; i < (size_t)(S->row_hi - need_lo) * X; ++i) nsrc += buf[i] != 0;
  }
  int rc = eu_set_source_count(S, nsrc);
  if (rc) return rc;
  if (!gather(9, 8, S->row_lo, S->row_hi)) return EULER_EINVAL;      // g_precon: the own rows, into the band-skewed array
  if ((rc = euler_set_field(S, EULER_F_PRECON, buf.data(), (size_t)(S->row_hi - S->row_lo) * X * 8))) return rc;
  if (dye)      // the window's rows (ghost rows included), like u and v
    for (int k = 0; k < 6; ++k) {
      if (!gather(10 + k, 4, need_lo, need_hi)) return EULER_EINVAL;
      HIPCHK(hipMemcpy(S->dye[k] + (size_t)need_lo * X, buf.data(), (size_t)rows * X * 4, hipMemcpyHostToDevice));
    }

  // ---- markers: those inside the own rows, with their keys (a whole-grid handle: all of them, ordered by key)
  std::vector<float> mk;
  std::vector<unsigned int> keys;
  uint64_t n_loc = 0;
  if (!S->slab_on) {
    if (h0.n_markers > S->max_markers) { eu_set_error("snapshot holds %llu markers, more than this handle's capacity", (unsigned long long)h0.n_markers); return EULER_EINVAL; }
    try { mk.resize(2 * (size_t)h0.n_markers + 2); } catch (...) { return EULER_ENOMEM; }
    std::vector<char> seen;
    try { seen.assign((size_t)h0.n_markers, 0); } catch (...) { return EULER_ENOMEM; }
    uint64_t placed = 0;
    for (Part* p : use) {
      const float* m = p->mk.data();
      const unsigned int* kk = p->keys.data();
      for (uint64_t i = 0; i < p->keys.size(); ++i) {
        const uint64_t key = kk[i];
        if (key >= h0.n_markers) { eu_set_error("%s: marker key %llu out of range", path, (unsigned long long)key); return EULER_EINVAL; }
        if (seen[(size_t)key]) { eu_set_error("%s: marker key %llu appears twice", path, (unsigned long long)key); return EULER_EINVAL; }
        seen[(size_t)key] = 1;
        mk[2 * key] = m[2 * i]; mk[2 * key + 1] = m[2 * i + 1];
        ++placed;
      }
    }
    if (placed != h0.n_markers) { eu_set_error("%s: the part files hold %llu of %llu markers", path, (unsigned long long)placed, (unsigned long long)h0.n_markers); return EULER_EINVAL; }
    if ((rc = euler_set_markers(S, mk.data(), h0.n_markers))) return rc;
    if ((rc = euler_set_rng(S, h0.rng_state, h0.source_exhausted))) return rc;
  } else {
    for (Part* p : use) {      // (read_part kept exactly the markers inside the own rows)
      mk.insert(mk.end(), p->mk.begin(), p->mk.end());
      keys.insert(keys.end(), p->keys.begin(), p->keys.end());
    }
    n_loc = keys.size();
    if (n_loc > S->max_markers) { eu_set_error("row slab %d: %llu markers in its rows, more than its capacity %zu", S->cfg.slab_rank, (unsigned long long)n_loc, S->max_markers); return EULER_ENOMEM; }
    if (n_loc) {
      HIPCHK(hipMemcpy(S->markers[S->cur], mk.data(), (size_t)n_loc * 8, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(S->keys[S->cur], keys.data(), (size_t)n_loc * 4, hipMemcpyHostToDevice));
    }
    MarkerState m0;
    memset(&m0, 0, sizeof m0);
    m0.n = h0.n_markers; m0.n_loc = n_loc; m0.max_markers = 4 * S->C; m0.rng_state = h0.rng_state; m0.exhausted = h0.source_exhausted;
    HIPCHK(hipMemcpy(S->ms, &m0, sizeof m0, hipMemcpyHostToDevice));
    S->n_markers_host = n_loc;
  }
  *h_out = h0;
  return EULER_OK;
}

// Collective on a row-slab handle: every rank loads its rows and markers on its own, then the ranks AGREE on the outcome (one all-reduce of the
// status) before the first collective of the restore - a missing or corrupt part file, a row no file holds, more markers than a slab's capacity or
// an allocation failure on ONE rank fails the call on EVERY rank (the handle is then unloaded everywhere) instead of leaving the others waiting.
extern "C" int euler_load_state(euler_sim* S, const char* path) {
  if (!S || !path) return EULER_EINVAL;
  if (S->slab_on && !S->has_comm) { eu_set_error("row-slab handle: install the communicator before loading a snapshot"); return EULER_ESTATE; }
  SnapHeader h0;
  memset(&h0, 0, sizeof h0);
  int rc = load_state_local(S, path, &h0);
  if (S->slab_on) {
    int worst = rc;
    const int rc2 = eu_slab_status_sync(S, rc, &worst);
    if (rc2) return rc2;
    if (worst) {
      if (!rc) eu_set_error("euler_load_state(%s): another rank could not load its part of the state", path);
      S->loaded = 0;
      return rc ? rc : worst;
    }
    // every rank compared the headers of the part files IT read; ranks whose rows lie in different files must have seen the same state too (an overwrite that died between
    // the renames of the parts leaves old and new files side by side)
    const double id[4] = {(double)h0.frames, (double)(h0.rng_state & 0xffffffffull), (double)(h0.rng_state >> 32), (double)h0.n_markers};
    int same = 1;
    if ((rc = eu_slab_same_everywhere(S, id, 4, &same))) return rc;
    if (!same) { eu_set_error("euler_load_state(%s): the ranks read part files of different states (an interrupted overwrite?)", path); S->loaded = 0; return EULER_EINVAL; }
    if ((rc = eu_slab_after_restore(S))) return rc;      // collective: the source cells of all ranks
    if ((rc = eu_sync_marker_state(S))) return rc;
  } else if (rc) return rc;
  S->lean_ok = 0;      // the solver arrays may hold another state's pressure and masks: the next assembly writes them whole
  eu_state_replaced(S);      // (the solid / sink grids were replaced)
  memset(&S->stats, 0, sizeof S->stats);
  S->stats.frames = h0.frames; S->stats.total_substeps = h0.total_substeps; S->stats.total_pcg_iterations = h0.total_pcg_iterations;
  S->loaded = 1;
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------ render on a row-slab handle
// draw_rows (main.c:914-951) reads g_solid, g_sink, g_marker_count in the visible rows [max(Y-1-wy, 1), Y-1).  Collective: every
// rank contributes the visible rows it owns (an all-gather per grid, in place in a staging buffer), and every rank formats the same
// frame.  A terminal-sized window is a few dozen rows of the top slab; the call works for any window.
int eu_slab_render(euler_sim* S, int wx, int wy, char* out, int cap, int* len) {
  if (!S->has_comm) { eu_set_error("row-slab handle without a communicator"); return EULER_ESTATE; }
  const int X = S->X, Y = S->Y, R = S->cfg.slab_nranks;
  int cutoff = Y - 1 - wy;
  if (cutoff < 1) cutoff = 1;
  const int vis_lo = cutoff, vis_hi = Y - 1 > cutoff ? Y - 1 : cutoff, vrows = vis_hi - vis_lo;
  const bool dye = S->dye[0] != nullptr;
  // planes: solid, sink, count (bytes) and, with --rainbow, g_r, g_g, g_b (floats)
  const int nplanes = dye ? 6 : 3;
  const void* grids[6] = {S->solid, S->sink, S->count, S->dye[0], S->dye[1], S->dye[2]};
  const size_t elem[6] = {1, 1, 1, 4, 4, 4};
  size_t poff[7] = {0};
  for (int k = 0; k < nplanes; ++k) poff[k + 1] = poff[k] + (size_t)vrows * X * elem[k];
  uint8_t* stage = nullptr;
  HIPCHK(hipMalloc((void**)&stage, poff[nplanes] + 16));
  std::vector<int64_t> off(R), cnt(R);
  int rc = EULER_OK;
  for (int k = 0; k < nplanes && !rc && vrows > 0; ++k) {
    for (int r = 0; r < R; ++r) {
      const int lo = 64 * S->part_lo[r], hi = 64 * S->part_hi[r] < Y ? 64 * S->part_hi[r] : Y;
      const int a = lo > vis_lo ? lo : vis_lo, b = hi < vis_hi ? hi : vis_hi;
      off[r] = (int64_t)poff[k] + (b > a ? (int64_t)(a - vis_lo) * X * (int64_t)elem[k] : 0);
      cnt[r] = b > a ? (int64_t)(b - a) * X * (int64_t)elem[k] : 0;
      if (r == S->cfg.slab_rank && b > a)
        (void)hipMemcpyAsync(stage + off[r], static_cast<const char*>(grids[k]) + (size_t)a * X * elem[k], (size_t)cnt[r], hipMemcpyDeviceToDevice, S->stream);
    }
    if (S->bulk.allgather(S->bulk.ctx, stage, off.data(), cnt.data()) != 0) { eu_set_error("communicator callback failed: render all-gather"); rc = EULER_ECOMM; }
  }
  const size_t C = S->C;
  uint8_t* g = rc ? nullptr : (uint8_t*)calloc(3, C);      // (untouched pages of the full-size planes are never committed)
  float* col = (rc || !dye) ? nullptr : (float*)calloc(3 * C, sizeof(float));
  if (!rc && (!g || (dye && !col))) rc = EULER_ENOMEM;
  if (!rc) {
    for (int k = 0; k < nplanes && vrows > 0; ++k) {
      void* dst = k < 3 ? (void*)(g + (size_t)k * C + (size_t)vis_lo * X) : (void*)(col + (size_t)(k - 3) * C + (size_t)vis_lo * X);
      if (hipMemcpyAsync(dst, stage + poff[k], poff[k + 1] - poff[k], hipMemcpyDeviceToHost, S->stream) != hipSuccess) rc = EULER_EHIP;
    }
    if (hipStreamSynchronize(S->stream) != hipSuccess) rc = EULER_EHIP;
  }
  if (!rc) rc = dye ? euler_render_grids_rgb(g, g + C, g + 2 * C, col, col + C, col + 2 * C, X, Y, wx, wy, out, cap, len)
                    : euler_render_grids(g, g + C, g + 2 * C, X, Y, wx, wy, out, cap, len);
  free(g);
  free(col);
  (void)hipFree(stage);
  return rc;
}
