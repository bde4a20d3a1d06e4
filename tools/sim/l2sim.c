// l2sim.c -- developer tool: trace-driven model of the 8 private XCD L2s for hop-kernel schedules (tools/sim/README).
// Input (binary, little endian): header int64 {n_wg, n_streams, n_cols, slots_per_xcd, quantum, ways, sets, persistent}
//   int64 wg_stream_ptr[n_wg+1]; int64 stream_ptr[n_streams+1]; int32 cols[n_cols]   (cols >= 0: X row id; < 0: streaming line)
// Model: workgroup b runs on XCD b % 8 (round-robin dispatch); each XCD has `slots` concurrently resident workgroups that take
// turns; a turn issues `quantum` accesses from every stream of the workgroup.  One cache "line" = one 256-byte X row.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int64_t wg; int64_t* pos; int nstreams; int64_t s0; } Slot;

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: l2sim trace.bin\n"); return 1; }
  FILE* f = fopen(argv[1], "rb");
  int64_t h[8];
  if (!f || fread(h, 8, 8, f) != 8) return 2;
  const int64_t n_wg = h[0], n_streams = h[1], n_cols = h[2], slots = h[3], quantum = h[4], ways = h[5], sets = h[6], rounds = h[7];
  int64_t* wgp = malloc((n_wg + 1) * 8); int64_t* sp = malloc((n_streams + 1) * 8); int32_t* cols = malloc(n_cols * 4 + 4);
  if (fread(wgp, 8, n_wg + 1, f) != (size_t)(n_wg + 1) || fread(sp, 8, n_streams + 1, f) != (size_t)(n_streams + 1) ||
      fread(cols, 4, n_cols, f) != (size_t)n_cols) return 3;
  fclose(f);
  int64_t hits_tot = 0, miss_tot = 0, shits = 0, smiss = 0;
  for (int xcd = 0; xcd < 8; ++xcd) {
    // cache: tags[set][way], lru stamps
    int32_t* tags = malloc(sets * ways * 4); int64_t* stamp = calloc(sets * ways, 8);
    memset(tags, 0xff, sets * ways * 4);
    int64_t clock = 0, hits = 0, miss = 0;
    Slot* sl = calloc(slots, sizeof(Slot));
    int64_t next = xcd;   // next workgroup id for this XCD
    int active = 0;
    for (int i = 0; i < slots; ++i) { sl[i].wg = -1; sl[i].pos = NULL; }
    int filling = 1, release = 0;
    for (;;) {
      int any = 0, blocked_any = 0, progress = 0, release_used = 0;
      for (int i = 0; i < slots; ++i) {
        Slot* s = &sl[i];
        if (s->wg < 0) {
          if (next >= n_wg) continue;
          if (rounds && active > 0 && !filling) continue;     // round barrier: refill only when the XCD has drained
          filling = 1;
          s->wg = next; next += 8;
          s->s0 = wgp[s->wg]; s->nstreams = (int)(wgp[s->wg + 1] - s->s0);
          s->pos = realloc(s->pos, (s->nstreams + 1) * 8);
          for (int k = 0; k < s->nstreams; ++k) s->pos[k] = sp[s->s0 + k];
          ++active;
        }
        any = 1;
        int live = 0;
        for (int k = 0; k < s->nstreams; ++k) {
          int64_t p = s->pos[k], e = sp[s->s0 + k + 1];
          for (int u = 0; u < quantum && p < e; ++u, ++p) {
            int32_t c = cols[p];
            if (c == INT32_MIN) { if (release) { release_used = 1; continue; } blocked_any = 1; break; }
            const int stream_line = c < 0;
            uint32_t key = (uint32_t)c;                                  // streaming lines use the negative id space
            uint64_t hsh = (uint64_t)key * 0x9E3779B97F4A7C15ull;
            int64_t set = (int64_t)((hsh >> 20) % (uint64_t)sets);
            int32_t* t = tags + set * ways; int64_t* st = stamp + set * ways;
            int w, victim = 0; int64_t oldest = INT64_MAX;
            for (w = 0; w < ways; ++w) { if (t[w] == c) break; if (st[w] < oldest) { oldest = st[w]; victim = w; } }
            ++clock;
            if (w < ways) { st[w] = clock; if (stream_line) ++shits; else ++hits; }
            else { t[victim] = c; st[victim] = clock; if (stream_line) ++smiss; else ++miss; }
          }
          s->pos[k] = p;
          if (p < e) { live = 1; if (cols[p] != INT32_MIN) progress = 1; }
        }
        if (!live) { s->wg = -1; --active; }
      }
      if (!any) break;
      filling = (active == 0);
      // all live streams of the XCD wait at a marker: open the barrier for one sweep
      release = (!release && blocked_any && !progress) ? 1 : 0;
      (void)release_used;
    }
    hits_tot += hits; miss_tot += miss;
    free(tags); free(stamp);
    for (int i = 0; i < slots; ++i) free(sl[i].pos);
    free(sl);
  }
  printf("gathers %lld  L2 hits %lld (%.1f%%)  misses %lld -> fetched %.2f GB of X rows (256 B each); stream lines: %lld hit %lld miss (%.2f GB)\n",
         (long long)(hits_tot + miss_tot), (long long)hits_tot, 100.0 * hits_tot / (hits_tot + miss_tot + 1e-9), (long long)miss_tot,
         miss_tot * 256.0 / 1e9, (long long)shits, (long long)smiss, smiss * 256.0 / 1e9);
  return 0;
}
