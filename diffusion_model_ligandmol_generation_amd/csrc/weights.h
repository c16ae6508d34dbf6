// Slot numbering of the weights table (host array of device pointers) and its mapping to
// the reference's state_dict keys (model1.py:595-655).
#pragma once
#include <stdint.h>

// The table has room for LD_MAXL E(3) layers and LD_MAXL attention layers (num_layers up to 2 LD_MAXL,
// model1.py:627-634 builds num_layers // 2 of each); a model with fewer leaves the slots of the others NULL and
// says how many it has in ligdiff_options.layer_pairs (default 3, the reference's num_layers = 6).
#define LD_MAXL 8

enum {
  W_ATOM_W = 0, W_ATOM_B, W_ATOM_G, W_ATOM_BE,          // atom_embedding.{0,1}
  W_POS_W, W_POS_B, W_POS_G, W_POS_BE,                  // pos_embedding.{0,1}
  W_T0_W, W_T0_B, W_T2_W, W_T2_B,                       // time_mlp.{0,2}
  W_NULL_EMB, W_PAR0_W, W_PAR0_B, W_PAR2_W, W_PAR2_B,   // property_guidance.*
  W_PENC0,                                              // 8 encoders x {0.w,0.b,2.w,2.b}
  W_E3_0 = W_PENC0 + 32,
  // per E(3) layer (20 slots)
  E3_RB0_W = 0, E3_RB0_B, E3_RB2_W, E3_RB2_B,
  E3_EN0_W, E3_EN0_B, E3_EN2_W, E3_EN2_B, E3_EN3_G, E3_EN3_B,
  E3_NN0_W, E3_NN0_B, E3_NN2_W, E3_NN2_B, E3_NN3_G, E3_NN3_B,
  E3_CN0_W, E3_CN0_B, E3_CN2_W, E3_CN2_B,
  E3_SLOTS,
  W_DIT_0 = W_E3_0 + LD_MAXL * E3_SLOTS,
  // per DiT layer (10 slots)
  DIT_ADA_W = 0, DIT_ADA_B, DIT_INP_W, DIT_INP_B, DIT_OUT_W, DIT_OUT_B,
  DIT_FF0_W, DIT_FF0_B, DIT_FF3_W, DIT_FF3_B,
  DIT_SLOTS,
  W_HEAD_0 = W_DIT_0 + LD_MAXL * DIT_SLOTS,
  // per head (atom, pos, property): {0.w, 0.b, 2.w(LN), 2.b(LN), 3.w, 3.b}
  HEAD_W0 = 0, HEAD_B0, HEAD_G, HEAD_BE, HEAD_W3, HEAD_B3,
  HEAD_SLOTS,
  W_TIME_FREQS = W_HEAD_0 + 3 * HEAD_SLOTS,  // not a state_dict entry: 128 Fourier frequencies
  // moldiff_consistency.* (model1.py:383-414): three predictors x {0.w, 0.b, 2.w(LN), 2.b(LN), 3.w, 3.b,
  // last.w, last.b} (last = layer 6 for atom / bond, layer 5 for valency) + the two chemistry buffers
  W_CONS_0,
  CONS_W0 = 0, CONS_B0, CONS_G, CONS_BE, CONS_W3, CONS_B3, CONS_WL, CONS_BL,
  CONS_SLOTS,
  W_CONS_MAXVAL = W_CONS_0 + 3 * CONS_SLOTS, W_CONS_RADII,
  W_COUNT
};
enum { CONS_ATOM = 0, CONS_BOND = 1, CONS_VAL = 2 };
#define W_CONS(p, s) (W_CONS_0 + (p) * CONS_SLOTS + (s))

#define W_E3(l, s) (W_E3_0 + (l) * E3_SLOTS + (s))
#define W_DIT(l, s) (W_DIT_0 + (l) * DIT_SLOTS + (s))
#define W_HEAD(h, s) (W_HEAD_0 + (h) * HEAD_SLOTS + (s))
#define W_PENC(i, s) (W_PENC0 + (i) * 4 + (s))

struct WeightInfo { char name[96]; int64_t numel; };
const WeightInfo* weight_table();  // W_COUNT entries
// Layer a slot belongs to (E(3) or attention layer index), -1 for the slots every model has.
static inline int weight_slot_layer(int slot) {
  if (slot >= W_E3_0 && slot < W_DIT_0) return (slot - W_E3_0) / E3_SLOTS;
  if (slot >= W_DIT_0 && slot < W_HEAD_0) return (slot - W_DIT_0) / DIT_SLOTS;
  return -1;
}
// Layer pairs of the calling thread's current call (ligdiff_options.layer_pairs; 3 when not set).
int call_layer_pairs();
