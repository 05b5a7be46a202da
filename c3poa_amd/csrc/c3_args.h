// c3_args.h -- kernel argument blocks shared by the launchers (c3_api.hip) and the kernels.
#pragma once
#include "c3_dev.h"

struct ConkArgs {
  C3Batch b; const uint8_t* sp_codes; const int* sp_len; int32_t* track; C3Info* info; int* counter;
  int match, mismatch, penalty;
  int n_spl; int32_t* scan;      // scan mode: [n * n_spl * 2][4] = max, argmax, mean, L
};
struct PeaksArgs {
  C3Batch b; const int32_t* track; C3Info* info; double* bufA; double* bufB; int32_t* cand; uint8_t* cstate;
  int32_t* raw_peaks; int32_t* n_raw; const int* sp_len; double coef[64]; int64_t maxL; int window, iters, min_dist;
  int* queue;                              // read queue (one int, zeroed before the launch)
};
struct PoaArgs {
  C3Batch b; C3Info* info; C3Params p; int* counter; const int* work; int n_work;
  // per-slot scratch, one block per kind (the kernel derives every array from these bases: few live SGPRs):
  //   ibase: 19*Ncap ints  (n_in n_out grp order order2 index gfirst glast rem mpl mpr rowm[3N] anchor col col2t nxt foff)
  //   ebase: 3*Ncap*K ints (in_from out_to out_w);  cellsb: 2*cells_cap bytes (direction bytes D8, predecessor bytes P8) + 16 * (cells_cap >> 2) bytes (first pass; >> 0 in the 32-bit pass) of 32-bit cells H E1 E2 D for the few rows that keep them;  bbase: 5*Ncap bytes (base rows2[4N])
  int* ibase; int* ebase; char* cellsb; uint8_t* bbase; long long* score;
  int Ncap, K, Pcap, cells_cap;
  uint8_t* draft; int32_t* tpos; uint8_t* msa_dbg; const int64_t* msa_off; int* msa_len;
  unsigned long long* phases; uint4* desc; int* jump;
  int* pbase;               // [slots][Pcap] node of every fused base
  int* overflow;            // reads whose scratch overflowed (count in counter[4]): redone with worst-case scratch; nullptr in the passes that have it
  int* overflow16;          // reads with a score beyond the 16-bit cells (count in counter[5]): redone by the 32-bit instance; nullptr in that pass
  int rb_span;              // 0 = default; test hook C3_DEBUG_POA_RBSPAN: width of the window a row maximum may move in before the 16-bit base follows it
};
struct WLayer { int qbeg, len, begin, end; };
struct WinRec { int rid, w, n_layers, blen, tgs, out_len, polished, pad_; };
struct PrepArgs {
  C3Batch b; C3Info* info; C3Params p; int* counter; const int* work; int n_work;
  const uint8_t* draft; int32_t* tpos; int32_t* eH; uint8_t* eD; int64_t ecap; int* lw_first; int* lw_last; int NLcap, NWcap;
  WinRec* wrec; WLayer* wlay; int* win_base; int* n_windows; int wcap;
};
struct WinArgs {
  C3Batch b; C3Params p; int* counter; int n_win; const WinRec* wrec_in; WinRec* wrec; const WLayer* wlay; int NLcap;
  const uint8_t* draft;
  uint8_t* base; int* ibase; int* ebase; long long* score;
  int32_t* H; uint16_t* D; uint4* rdesc; int Ncap, K; long long hcap; uint8_t* wout; int wout_cap;
  int Lcap;                               // nodes the LDS consensus sweep can hold (<= Ncap)
  unsigned long long* phases;
  int band_mode;                          // 0 = banded rows with certificate (default), 1 = never banded, 2 = every certificate counts as failed (test hook: C3_DEBUG_BAND)
  // two launches: the first with DP scratch for the typical layer (hcap small); a window one of whose layers does not fit is
  // dropped untouched into `ovf_list` (count in counter[W_CNT_OVF]) and redone by the second launch, which has worst-case scratch,
  // takes its windows from `wlist`, their number from device memory (`n_win_dev`) and its queue from counter[W_CNT_Q2]
  // (k_window<true>: the first launch's code carries none of this)
  const int* wlist; const int* n_win_dev; int* ovf_list;
  // round 6: done_flag != nullptr makes the second launch a CONSUMER that runs beside the first one on a stream of its own: a few resident waves
  // claim an entry of the list whenever it holds one they have not taken (CAS on counter[W_CNT_Q2] against counter[W_CNT_OVF]), sleep otherwise,
  // and stop when the flag is set (by the host, on the first launch's stream, behind that launch) and the list is empty.  A third, ordinary
  // launch (done_flag = nullptr) mops up what they left
  const int* done_flag;
  int dbg_ovf_delay;                      // test hook (C3_DEBUG_OVF_DELAY_MS): the first launch waits this long between taking an overflow index and writing the entry
  // output of the second launch: entry q of its queue writes wout2 + q * wout2_cap (a window consensus can be as long as the graph has
  // nodes; the first launch's slots hold 3 windows + 64 and hand longer ones over); WinRec::pad_ = q + 1 tells k_stitch where to look
  uint8_t* wout2; int wout2_cap, wout2_n;
};
#define W_CNT_OVF 48                      /* d_counter ints: [0] queue of the first launch, [48] overflow count, [49] queue of the second */
#define W_CNT_Q2 49
#define W_CNT_START 55                    /* [55] raised by every wave of the first launch: the consumer beside it only waits for a launch that runs */
#define W_CNT_DONE 54                     /* [54] set by the host behind the first launch: its overflow list is complete */
#define W_CNT_WHY 50                      /* [50..53] windows given up: backbone longer than the output slot, DP scratch, graph nodes, consensus length */
struct StitchArgs {
  C3Batch b; C3Info* info; const int* work; int n_work; const WinRec* wrec; const int* win_base; const uint8_t* wout; int wout_cap; char* cons;
  const uint8_t* zflag;
  const uint8_t* wout2; int wout2_cap;
};
// adapter finder (k_adapter): every read x every entry of the splint table x both strands
struct AdapterArgs {
  C3Batch b; C3Params p; int* counter;
  const uint8_t* ad_codes; const int* ad_len; int n_ad;
  uint8_t* D; long long dcap, dstride;  // [grid][dstride]: dcap direction bytes, then two rows of rowcap + 1 ints
  int rowcap;                           // (H and E of the previous row for front pieces beyond the LDS rows; 0 = none)
  int32_t* out;                         // [n * n_ad * 2][12]
};

struct ZeroArgs {
  C3Batch b; C3Info* info; C3Params p; int* counter; const int* work; int n_work;
  uint8_t* D; long long dcap, dstride;  // [grid][dstride]: dcap direction bytes, then two rows of rowcap + 1 ints
  int rowcap;                           // (H and E of the previous row for front pieces beyond the LDS rows; 0 = none)
  int4* zinfo; uint8_t* zflag;          // per read: r_st, r_en, q_st, q_en; rescue in progress / done
  const uint8_t* draft; char* cons;
};
