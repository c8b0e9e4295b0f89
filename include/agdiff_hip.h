/*
 * agdiff_hip.h -- C ABI of libagdiff_hip.so: the MI355X (gfx950) kernels behind the AGDIFF
 * diffusion-sampling hot path.
 *
 * The reference (ADicksonLab/AGDIFF) is pure Python; the native work on this path is done by
 * third-party packages it calls (torch_cluster / torch_scatter / torch_sparse / PyG) and by
 * ATen.  Each entry point below names the reference call site(s) it replaces
 * (paths relative to the reference tree).  All pointers are DEVICE pointers unless marked
 * [host]; all tensors are dense, row-major; indices are int32 on this side of the boundary
 * (the Python host converts from/to the reference's int64).  Every function enqueues work on
 * `stream` (a hipStream_t passed as void*) and returns 0, or a negative agdiff_status code
 * after validating its arguments on the host.  Nothing here allocates, frees or synchronises.
 *
 * Struct fields are only `void*`-sized pointers, int64_t, int32_t and float so that the
 * Python binding (agdiff_amd/_lib.py) can mirror them mechanically by parsing this header.
 */
#ifndef AGDIFF_HIP_H
#define AGDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGDIFF_ABI_VERSION 46
#define AGDIFF_HIDDEN 128          /* config.hidden_dim; InteractionBlock.lin hard-codes 256 = 2*128 (schnet.py:190) */
#define AGDIFF_MAX_CONVS 8         /* >= config.num_convs (6) */
#define AGDIFF_MAX_CONVS_LOCAL 8   /* >= config.num_convs_local (4) */
#define AGDIFF_NUM_EDGE_TYPES 100  /* rows of bond_emb (edge.py:49) */
#define AGDIFF_MAX_ATOMS_PER_GRAPH 512
#define AGDIFF_RADIUS_CAP 33       /* max_num_neighbors + 1 (torch_cluster.radius_graph, common.py:217) */
#define AGDIFF_TILE 16             /* edges / nodes per MFMA tile */
#define AGDIFF_MAX_CHUNK_TILES 8   /* most tiles one wave walks per chunk in the fused CFConv kernel (128 edges) */
#define AGDIFF_RMSD_MAX_ATOMS 256  /* most (heavy) atoms per conformer in agdiff_rmsd_matrix */
#define AGDIFF_POLY_MAX_KT 4       /* most 32-term k-tiles of the radius-edge filter polynomial (degree 127): 1, 2 what smooth
                                      checkpoints take; 3, 4 the rungs between them and the filter MLPs for sharp ones */
#define AGDIFF_POLY_MAX_SLOTS 16   /* most local edge types with filter polynomials; the first sets that fit stay in LDS next to the
                                      radius edges' set (5 at poly_kt 1), the others are read from L2 by the tiles that meet them */
#define AGDIFF_RAD_STRIDE 48       /* rows reserved per target in the radius-edge list (three 16-row tiles >= AGDIFF_RADIUS_CAP) */

enum agdiff_status {
  AGDIFF_OK = 0,
  AGDIFF_ERR_ARG = -1,       /* null pointer / negative size / inconsistent sizes */
  AGDIFF_ERR_LIMIT = -2,     /* a compile-time limit above is exceeded */
  AGDIFF_ERR_LAUNCH = -3     /* hipGetLastError() after a launch */
};

/* ---- packed network weights (built by agdiff_amd/packing.py from the reference state_dict) ---
 * "pk" = MFMA-operand-major packing of a Linear weight W[out][in] in 2-KiB blocks of 16 outputs x 32 inputs
 * (OT = ceil(out/16), KT = ceil(in/32), zero padded), blocks ordered [OT][KT]; "pkk" = the same blocks in
 * k-tile-outer order [KT][OT], for layers whose input is streamed in 32-feature k-tiles.  Lane l of block
 * (ot, t) holds the eight weights W[16*ot + (l&15)][32*t + c], c in {4q..4q+3, 16+4q..16+4q+3}, q = l>>4:
 *   precision 0: two 16-byte units [u][lane][4 fp32] (elements 4u..4u+3)
 *   precision 1: two 16-byte units [part][lane][8 bf16], part 0 = bf16(w), part 1 = bf16(w - hi)
 *   precision 2: the same with fp16 (agdiff_params_t.precision_local)
 * Dimensions in the field comments below are [OT][KT] block counts.  Vectors are in natural feature order. */
typedef struct agdiff_conv_params {
  /* CFConv filter networks of one InteractionBlock (schnet.py:169-186), conv1 (F=128) and conv2 (F=64) fused.
   * ShiftedSoftplus (schnet.py:71-80) is evaluated in base 2 with its constants folded into the two linear
   * layers by the host: with c = beta * log2(e) per conv, layer 1 yields u = c (W1 a + b1); the kernel forms
   * s = max(u, log2(1 + 2^u)); since softplus(beta x) - ln 2 = ln 2 (s - 1), layer 2 uses ln 2 * W2 and
   * b2 - ln 2 * W2 1. */
  const float* filt_w1_pk;   /* pkk [4][12]: rows 0..127 c1 * conv1.nn.0.weight, 128..191 c2 * conv2.nn.0.weight */
  const float* filt_b1;      /* [192] c * nn.0.bias */
  const float* filt_w2a_pk;  /* pk [8][4]: ln2 * conv1.nn.2.weight */
  const float* filt_w2b_pk;  /* pk [4][2]: ln2 * conv2.nn.2.weight */
  const float* filt_b2;      /* [192] nn.2.bias - ln2 * rowsum(nn.2.weight) */
  const float* dist_seg;     /* [2][100]: DistanceWeightingNetwork (schnet.py:83-100) of conv1 / conv2 by segments: the network's
                                pre-sigmoid value layer2(relu(layer1(d))) is piecewise linear in d with at most 32 kinks
                                (d = -b1_k / w1_k); bp[32] = the kinks in ascending order (padded with +inf), then
                                alpha[33], beta[33] with value = alpha[s] d + beta[s] on segment s = #{kinks <= d}, summed
                                over the active hidden units in float64 by the host; [98..99] unused */
  /* node side of the block (schnet.py:153-158, 201-216, 219-234) */
  const float* lin1_pk;      /* pk [12][4]: BN-folded conv1.lin1 (rows 0..127) and conv2.lin1 (128..191) */
  const float* lin1_b;       /* [192] */
  /* InteractionBlock.act (ShiftedSoftplus with learnable beta, schnet.py:71-80,206) is evaluated in base 2 like the filter
   * networks' one: kb = act.beta * log2(e) is folded into lin2 (u = kb (W2 agg + b2)), the kernel forms
   * s = max(u, log2(1 + 2^u)), and lin takes ln2 * W with bias b - ln2 * W 1  (softplus(beta x) - ln 2 = ln 2 (s - 1)) */
  const float* lin2a_pk;     /* pkk [4][8]: kb * BN-folded conv1.lin2 */
  const float* lin2b_pk;     /* pkk [2][8]: kb * BN-folded conv2.lin2 */
  const float* lin2_b;       /* [256] kb * (BN-folded biases) */
  const float* lin_pk;       /* pk [8][8]: ln2 * InteractionBlock.lin (256->128) */
  const float* lin_b;        /* [128] lin.bias - ln2 * rowsum(lin.weight) */
  const float* gate1_pk;     /* pk [4][4]: attention.0 (128->64) */
  const float* gate1_b;      /* [64] */
  const float* gate2_w;      /* [64]  attention.2 */
  const float* scale1_pk;    /* pk [1][4]: scaling fc.0 (128->8, rows 8..15 zero) */
  const float* scale2_pk;    /* pk [8][1]: scaling fc.2 (8->128, cols 8..31 zero) */
  const float* filt_poly_pk; /* pk [12][poly_kt] or null: the whole filter network of a RADIUS edge (type 0, d < cutoff) as a
                                polynomial in d -- see agdiff_params_t.poly_kt.  Rows 0..127 conv1, 128..191 conv2; nn.2.bias
                                is the constant term */
  const float* filt_poly_typed_pk; /* [poly_num_slots] x pk [12][poly_kt] or null: the same for the LOCAL edge types that have
                                a slot (agdiff_params_t.poly_type_slot), d in [0, cutoff] (beyond it the CFConv's cutoff factor
                                C(d) is exactly 0, schnet.py:140-146) */
  float gate2_b;
  float act_beta;            /* InteractionBlock.act.beta (informational: folded into lin2 / lin above) */
  float filt_poly_unscale;   /* 2^-S (1 when unused): filt_poly_pk and filt_poly_typed_pk hold 2^S times the coefficients and
                                agdiff_cfconv_node multiplies a target's finished sums by 2^-S -- exact, and it lifts the lo
                                parts of small split-fp16 coefficients out of fp16's subnormal range (quantum 6e-8, i.e. up
                                to 1e-6 of a filter of size 0.2 over 32 terms); the host picks S with max |c| 2^S <= 1024 */
  float pad0;
} agdiff_conv_params_t;

typedef struct agdiff_gin_params {
  const float* w1_pk;        /* pk [8][4] convs.k.nn.layers.0 */
  const float* b1;           /* [128] */
  const float* w2_pk;        /* pk [8][4] BN-folded convs.k.nn.layers.1 */
  const float* b2;           /* [128] */
  float one_plus_eps;
  int32_t relu_out;          /* 1 for all but the last layer (gin.py:134) */
} agdiff_gin_params_t;

/* agdiff_head_params_t.act: torch.nn.functional names the heads' MultiLayerPerceptron may be built with (models/common.py:62-66) */
#define AGDIFF_ACT_RELU 0
#define AGDIFF_ACT_GELU 1        /* erf form (F.gelu default) */
#define AGDIFF_ACT_SILU 2
#define AGDIFF_ACT_TANH 3
#define AGDIFF_ACT_SIGMOID 4
#define AGDIFF_ACT_SOFTPLUS 5    /* beta 1, threshold 20 */
#define AGDIFF_ACT_LEAKY_RELU 6  /* negative_slope 0.01 */
#define AGDIFF_ACT_ELU 7         /* alpha 1 (F.celu with its default alpha is the same function) */
#define AGDIFF_ACT_RELU6 8
#define AGDIFF_ACT_HARDTANH 9    /* [-1, 1] */
#define AGDIFF_ACT_SELU 10
#define AGDIFF_ACT_MISH 11
#define AGDIFF_ACT_HARDSWISH 12
#define AGDIFF_ACT_HARDSIGMOID 13
#define AGDIFF_ACT_SOFTSIGN 14
#define AGDIFF_ACT_LOGSIGMOID 15
#define AGDIFF_ACT_HARDSHRINK 16  /* lambd 0.5 */
#define AGDIFF_ACT_SOFTSHRINK 17  /* lambd 0.5 */
#define AGDIFF_ACT_RRELU 18       /* as evaluated (training=False): negative slope (1/8 + 1/3) / 2 */
typedef struct agdiff_head_params {
  const float* w1_pk;        /* pkk [8][8] layers.0 (256->128) */
  const float* b1;           /* [128] */
  const float* w2_pk;        /* pk [4][4] layers.1 (128->64) */
  const float* b2;           /* [64] */
  const float* w3;           /* [64]  layers.2 */
  const float* attr_poly_pk; /* pkk [poly_kt][8] or null: layers.0.weight[:, 128:] @ edge_attr(d, type 0) as a polynomial in d
                                (agdiff_params_t.poly_kt) */
  float b3;
  int32_t act;               /* AGDIFF_ACT_*: config.mlp_act, the activation between the head's layers (models/common.py:62-66:
                                getattr(F, name); configs/*.yml: relu) */
  int32_t precision;         /* as agdiff_params_t.precision, or 2 (split-fp16: only with attr_rows) */
  int32_t pad0;
  int32_t* range_rows;       /* as agdiff_ws_t.range_rows (the head's hidden layer; an edge flags its source node), or null.
                                agdiff_score_forward passes the workspace's */
} agdiff_head_params_t;

typedef struct agdiff_params {
  /* MLPEdgeEncoder (edge.py:84-103), attention dropped (softmax over a size-1 axis == 1) and
   * edge_feature_mlp.2 folded into combination_mlp.0 */
  const float* ee_fe_w;      /* [128] feature_expansion.weight[:,0] */
  const float* ee_fe_b;      /* [128] */
  const float* ee_t1;        /* [100][128]: edge_feature_mlp.0.weight[:,128:] @ bond_emb[t] + bias */
  const float* ee_w1_pk;     /* pk [8][4]: edge_feature_mlp.0.weight[:,:128] */
  const float* ee_t3;        /* [100][128]: comb.0.weight[:,128:] @ bond_emb[t] + comb.0.bias + comb.0.weight[:,:128] @ efm.2.bias */
  const float* ee_w23_pk;    /* pk [8][4]: comb.0.weight[:,:128] @ edge_feature_mlp.2.weight */
  const float* ee_w4_pk;     /* pk [8][4]: combination_mlp.2 */
  const float* ee_b4;        /* [128] */
  /* GaussianSmearingEdgeEncoder (edge.py:17-42 with GaussianSmearing, schnet.py:18-27): used instead of the
   * ee_* fields when edge_encoder == 1; out = [exp(coeff (d - offset_k)^2), k < 64 | bond_emb[type]] */
  const float* ge_offset;    /* [64] rbf.offset buffer = linspace(0, 2 cutoff, 64) */
  const float* ge_emb;       /* [100][64] bond_emb.weight */
  const float* schnet_emb;   /* [100][128] encoder_global.embedding (max_norm renorm applied to used rows) */
  const float* gin_emb;      /* [100][128] encoder_local.node_emb */
  const int32_t* poly_type_slot; /* [100] or null: slot of an edge type in filt_poly_typed_pk, -1 = none */
  const float* dist_union;   /* [K + S * 2 * 2 num_convs] or null (K = dist_union_kinks, S = dist_union_segments below): the
                                DistanceWeightingNetworks of all CFConvs (conv[k].dist_seg) over their COMMON segments on [0, cutoff]:
                                [0..K-1] the union of their kinks in (0, cutoff] ascending (+inf padded; K a power of two > their
                                number, at most 512), then for union segment u = number of those kinks <= d and scale
                                cc = 2 k + (0: conv1, 1: conv2) the line (alpha, beta) at [K + (u * 2 num_convs + cc) * 2]: the same
                                floats dist_seg selects for a d in [0, cutoff], found with ONE search instead of one per conv
                                (agdiff_sampler_front; beyond the cutoff the envelope makes every scale exactly 0) */
  const float* attr_poly_typed_pk; /* [poly_num_slots + attr_poly_far_slots] x pk [8][1] or null: edge_attr itself (128 features) of a
                                local edge of a slotted type as a polynomial in d on [0, cutoff] (agdiff_local_edge_rows); then, for
                                the first attr_poly_far_slots slots, the same on [cutoff, attr_poly_far_hi] (below) */
  agdiff_conv_params_t conv[AGDIFF_MAX_CONVS];
  agdiff_gin_params_t gin[AGDIFF_MAX_CONVS_LOCAL];
  agdiff_head_params_t head_global;
  agdiff_head_params_t head_local;
  int32_t num_convs;
  int32_t num_convs_local;
  float cutoff;
  int32_t smooth;            /* config.smooth_conv */
  int32_t precision;         /* 0: exact fp32 MFMA; 1: split-bf16 (hi+lo, 3 MFMA passes, fp32 accumulate); 2: split-fp16 (the same scheme
                                on v_mfma_f32_16x16x32_f16: 11 + 11 mantissa bits per operand at the same rate; operands beyond
                                fp16's range saturate) -- selects both the kernels and the layout of every packed matrix and
                                of e_attr / l_attr */
  int32_t edge_encoder;      /* config.edge_encoder: 0 'mlp' (edge.py:45-103), 1 'gaussian' (edge.py:17-42) */
  float ge_coeff;            /* -0.5 / (offset[1] - offset[0])^2 (schnet.py:22) */
  int32_t poly_num_slots;    /* 0, or 1..AGDIFF_POLY_MAX_SLOTS: local edge types whose CFConv filters are d-polynomials too
                                (the types the host has met in batches so far and whose fit it accepted); 0 sends the local
                                edges through the filter MLPs.  A batch that brings a type without a slot (fit refused, or more
                                than AGDIFF_POLY_MAX_SLOTS types) runs MIXED: the slotted types' edges by polynomials inside
                                agdiff_cfconv_node, the others through agdiff_cfconv_local (topo->local_type_mask tells) */
  int32_t precision_local;   /* arithmetic of the LOCAL branch's MFMA kernels (GIN layers, local head, local edge_attr rows by
                                polynomial): 0 / 1 as `precision`, 2: split-fp16 (hi + lo fp16, three passes of
                                v_mfma_f32_16x16x32_f16: ~2^-20 per product (both parts truncated: one-sided) at the split-bf16 rate) -- gin[].w*_pk, head_local.w*_pk
                                and attr_poly_typed_pk are packed in THIS mode; the encoder MLP of flagged tiles stays in
                                `precision` */
  int32_t poly_kt;           /* 0: off.  1..AGDIFF_POLY_MAX_KT: radius edges (type 0: no bond embedding; d < cutoff by
                                construction) take their CFConv filters and the edge_attr half of the global head's first layer
                                from polynomials in d instead of the encoder + filter MLPs: both are smooth functions of the ONE
                                scalar d (edge.py:84-103 -> schnet.py:169-179), fitted by the host in float64 at Chebyshev nodes
                                and accepted only when they reproduce the networks to <= 1e-6 of the largest value on a dense
                                grid (agdiff_amd/packing.py).  K = 32 poly_kt terms in the product basis
                                  phi[8 g + j](x) = T_{8 g}(x) T_j(x),  x = 2 d / cutoff - 1,  g < 4 poly_kt,  j < 8
                                (T_n = Chebyshev polynomials; spans all polynomials of degree < K); operand element j of lane
                                quarter q in k-tile t is phi[8 (4 t + q) + j], and the packed blocks are ordered to match. */
  int32_t poly_plan;         /* passes of the split arithmetic over the filter polynomials' terms in agdiff_cfconv_node (0 when
                                precision == 0).  0: every term three passes (hi hi, lo hi, hi lo).  1: the HIGH terms -- f >= 16
                                at poly_kt 1, f >= 32 at poly_kt 2..4 -- take ONE pass (hi x hi): the host sets it only when
                                  fit error + eps1 * max_out sum_{high f} |c[out][f]| <= 1e-6 of the largest filter value
                                for the radius set and the common local types (|phi_f| <= 1; eps1 = 1.5 * 2^-10 split-fp16, 2^-7 split-bf16: the
                                two operand roundings of a single product), agdiff_amd/packing.py poly_pass_plan.
                                  poly_kt 1: TWO MFMAs per 16-channel tile instead of three -- unit 1 of every block of
                                  conv[].filt_poly_pk / filt_poly_typed_pk then holds, for lanes 0..31, their hi elements again
                                  and, for lanes 32..63, the LO elements of lane - 32 (same row, terms 8 (q - 2) + j), and the
                                  kernel's second operand is [lo(phi_f), f < 16 | hi(phi_f), f < 16]: both cross terms of the
                                  low 16 terms in one K = 32 instruction.
                                  poly_kt >= 2: k-tile 0 three passes, the others one (unit 1 of their blocks is not read);
                                  the high terms are then f >= 32.
                                2 (poly_kt >= 3) / 3 (poly_kt 4): k-tiles 0..1 / 0..2 three passes, the others one -- the high terms
                                are f >= 64 / 96: sharp networks whose terms 32..63 (..95) still carry too much weight for plan 1. */
  int32_t tune_cfconv_quad_tiles;    /* [0] agdiff_cfconv_node on quads (topo->group_targets == 4): radius rows in quad tiles too -- quarter k
                                        of a tile = four rows of the quad's k-th target, one set of sums per lane, no exchange between
                                        the quarters (k_cfconv_quad); -1: every target its own radius tiles (k_cfconv_node) */
  /* Kernel-variant thresholds: batch-size crossovers measured on MI355X.  0 selects the library default in
   * brackets; tests set them to reach every variant on small fixtures, agdiff_ws_t.variant_log reports what ran. */
  int64_t poly_slot_mask[2]; /* bit t of the 128-bit mask: edge type t has a slot in poly_type_slot */
  int64_t tune_share_rows_min_nodes;  /* [8192] batches with at least this many atoms take the local edges' edge_attr rows from
                                         the global encoder pass (ws->e_loc) instead of a pass over the canonical local list */
  int64_t tune_node_ldsw_min_tiles;   /* [1536] node stage / GIN layer: from this many 16-node tiles on, workgroups share the
                                         weights through LDS; below, every wave streams them from L2 */
  int64_t tune_node_split_max_tiles;  /* [320] node stage: up to this many tiles four waves share one tile (-1: never) */
  int32_t tune_serial_branches;       /* [0] 1: local and global branch on the caller's stream, no side stream */
  int32_t tune_local_poly_off;        /* [0] 1: local edges through the filter MLPs even when every type has a polynomial */
  int32_t tune_attr_poly_off;         /* [0] 1: agdiff_local_edge_rows evaluates the encoder MLP for every tile */
  int32_t tune_poly_lds_sets;         /* [0 = as many as fit in 160 KiB] agdiff_cfconv_node: at most this many coefficient sets in
                                         LDS, the radius edges' one included (1: every local type's set is read from L2) */
  int32_t attr_poly_far_slots; /* 0, or the number of FAR sets that follow the poly_num_slots near sets in attr_poly_typed_pk: edge_attr
                                  of a local edge type on [cutoff, attr_poly_far_hi].  Local edges LONGER than the cutoff -- bonded
                                  atoms far apart at high sigma, every local edge of a local-only step early in the schedule --
                                  then take their rows from it instead of the encoder MLP (beyond the cutoff the encoder's GELUs are
                                  saturated: a 32-term fit on [rc, 10 rc] is good to ~1e-9; accepted at <= 1e-6 like every fit).
                                  attr_poly_far_set[type] = index of the type's far set in attr_poly_typed_pk, or -1; the kernel's
                                  LDS limits poly_num_slots + attr_poly_far_slots to 9 (the host gives the 2-/3-hop types and the
                                  single bonds their far sets first) */
  const int32_t* attr_poly_far_set; /* [100] or null */
  float attr_poly_far_hi;      /* upper end of that range (agdiff_amd/packing.py: 10 x cutoff); longer edges keep the encoder MLP */
  int64_t tune_cfconv_four_min_quads; /* [8192] agdiff_cfconv_node at poly_kt 1: from this many quads on (two per wave of 256 x 16),
                                         16-wave workgroups at 128 VGPRs = four waves per SIMD, groups of two channel tiles;
                                         below, the 12-wave shape (more workgroups for the same quads); -1: never */
  int32_t dist_union_kinks;    /* K: kink slots at the head of dist_union (a power of two in [2, 512]) */
  int32_t dist_union_segments; /* S: segments (= kinks in (0, cutoff] + 1 <= K) whose lines follow */
} agdiff_params_t;

/* bits of agdiff_ws_t.variant_log: which kernel variants the launchers chose since the host last cleared it */
#define AGDIFF_VAR_CFCONV_NODE 1        /* agdiff_cfconv_node: radius rows (+ local quad tiles) by filter polynomials */
#define AGDIFF_VAR_CFCONV_NODE_LOCAL 2  /* ... its local quad tiles ran (every local type has a polynomial) */
#define AGDIFF_VAR_CFCONV_LOCAL_MLP 4   /* agdiff_cfconv_local: local edges through the filter MLPs */
#define AGDIFF_VAR_CFCONV_FUSED 8       /* agdiff_cfconv_fused: every edge through the filter MLPs */
#define AGDIFF_VAR_NODE_LDSW 16         /* node stage with workgroup-shared LDS weights */
#define AGDIFF_VAR_NODE_STREAM 32       /* node stage, one wave per tile streaming from L2 */
#define AGDIFF_VAR_NODE_SPLIT4 64       /* node stage, four waves per tile */
#define AGDIFF_VAR_GIN_LDSW 128         /* GIN layer with workgroup-shared LDS weights */
#define AGDIFF_VAR_SHARE_ROWS 256       /* local edge_attr rows written by the global encoder pass */
#define AGDIFF_VAR_ATTR_POLY 512        /* local edge_attr rows from per-type polynomials (agdiff_local_edge_rows) */
#define AGDIFF_VAR_HEAD_POLY 1024       /* global head with the edge_attr half from the d-polynomial */
#define AGDIFF_VAR_SIDE_STREAM 2048     /* local branch forked onto the side stream */
#define AGDIFF_VAR_POLY_L2_SETS 4096    /* agdiff_cfconv_node: some local types' coefficient sets did not fit in LDS (read from L2) */
#define AGDIFF_VAR_FUSED_FRONT 8192     /* agdiff_sampler_front: update of step t + radius graph of step t + 1 in one launch */
#define AGDIFF_VAR_CFCONV_NODE_FOUR 16384 /* agdiff_cfconv_node ran its four-waves-per-SIMD shape (tune_cfconv_four_min_quads) */
#define AGDIFF_VAR_CFCONV_NODE_QUAD 32768 /* agdiff_cfconv_node walked the radius rows in quad tiles (tune_cfconv_quad_tiles) */

/* ---- static topology of one packed batch (host builds it once per batch) ---------------------
 * Graphs are contiguous node ranges (PyG Batch, utils/misc.py:88-90).  "Local" edges are the
 * bond / 2-hop / 3-hop edges (type > 0, dualenc.py:566); they never depend on positions. */
typedef struct agdiff_topo {
  int64_t num_nodes;         /* N */
  int64_t num_graphs;        /* G */
  int64_t num_local;         /* L: local edges, in reference order (sorted by (src, dst)) */
  int64_t max_edges;         /* capacity of the per-edge buffers: sum_i (33 + local in-degree_i) */
  int64_t max_atoms_per_graph; /* <= AGDIFF_MAX_ATOMS_PER_GRAPH */
  int64_t max_in_degree;     /* max_i (33 + local in-degree_i); any value (a list may span several chunks) */
  const int32_t* graph_ptr;  /* [G+1] node offsets */
  const int32_t* atom_type;  /* [N] */
  const int32_t* loc_src;    /* [L] */
  const int32_t* loc_dst;    /* [L] */
  const int32_t* loc_type;   /* [L] (1..99) */
  const int32_t* loc_out_ptr;/* [N+1]: local edges with src == i are [loc_out_ptr[i], loc_out_ptr[i+1]) */
  const int32_t* loc_in_ptr; /* [N+1] */
  const int32_t* loc_in_eid; /* [L]: local edge ids grouped by dst (src ascending) */
  /* canonical local edges: the local edges j -> i and i -> j carry the same type and length, hence the same edge_attr and
   * the same local-head output (h_i * h_j is symmetric); one of the two (src < dst) is canonical, as is every local edge
   * without such a mirror.  Static like the local list itself. */
  int64_t num_local_canon;   /* Lc */
  const int32_t* lc_src;     /* [Lc] */
  const int32_t* lc_dst;     /* [Lc] */
  const int32_t* lc_type;    /* [Lc] */
  const int32_t* lc_pos;     /* [Lc]: the canonical edge's own id in the local list */
  const int32_t* lc_mir;     /* [Lc]: its mirror's id there, or -1 */
  const int32_t* loc_row;    /* [L]: canonical index (row of l_attr_rows) of every local edge */
  const int32_t* loc_in_src; /* [L]: loc_src[loc_in_eid[s]] (the GIN gather reads its indices by in-slot, one level deep) */
  const int32_t* loc_in_row; /* [L]: loc_row[loc_in_eid[s]] */
  /* the local list as a destination-sorted edge list of its own for the split CFConv, every non-empty target's list PADDED to
   * at least 8 entries: a 16-edge tile then holds at most three targets with the middle one's list whole, which the
   * kernels' fast reduction covers (local in-degrees are ~8: unpadded, many tiles would hold four or more and take the
   * general reduction).  Pad entries: src = dst = the target, type of the list's first edge, and nothing ever writes their
   * CFConv scale (zero-initialised): they contribute exactly 0. */
  int64_t num_local_padded;  /* Lp */
  const int32_t* lp_ptr;     /* [N+1]: padded list of target i = [lp_ptr[i], lp_ptr[i+1]) */
  const int32_t* lp_src;     /* [Lp] */
  const int32_t* lp_dst;     /* [Lp] */
  const int32_t* lp_type;    /* [Lp] */
  const int32_t* lp_row;     /* [Lp]: canonical index (row of l_attr_rows) of the entry's edge, -1 for pad entries */
  const int32_t* lc_ppos;    /* [Lc]: padded-list position of the canonical edge */
  const int32_t* lc_pmir;    /* [Lc]: ... of its mirror, or -1 */
  /* QUADS of targets for agdiff_cfconv_node: one wave owns the four targets quad_tgt[4 p .. 4 p + 3] of ONE molecule (-1: none,
   * in the last quad of a molecule).  The local edges once more as quad tiles: a 16-row tile holds rows of ONE edge type;
   * rows 4 k .. 4 k + 3 are in-edges of that type of the quad's k-th target (in order of source; a target with more than
   * four gets another tile of the type), pad rows: src = the target itself, the tile's type, and nothing ever writes their
   * CFConv scale: they contribute exactly 0.  The four rows one lane quarter holds then belong to ONE target, so the sum
   * over a target's edges needs no masks, and a tile needs ONE filter set.  The host groups atoms whose in-lists need like
   * numbers of tiles per type; tiles of quad p: type ascending */
  int64_t num_quads;         /* Q */
  int64_t group_targets;     /* GT = 4, 2 or 1: targets per quad that are used (small batches: fewer targets per wave, more waves).
                                A tile then gives each target 16 / GT rows: rows (16 / GT) k .. of the k-th target; entries
                                quad_tgt[4 p + GT ..] are -1 */
  const int32_t* quad_tgt;   /* [4 Q] */
  const int32_t* lcm_ptr;    /* [G+1]: the canonical local edges lc_*[lcm_ptr[g] .. lcm_ptr[g+1]) belong to molecule g (the list is
                                sorted by molecule, then edge type, then source) */
  int64_t local_type_mask[2];/* bit t of the 128-bit mask: the batch has a local edge of type t */
  int64_t num_local_tiles;   /* T = lt_ptr[Q] */
  const int32_t* lt_ptr;     /* [Q + 1]: tiles of quad p are [lt_ptr[p], lt_ptr[p+1]) */
  const int32_t* lt_src;     /* [16 T] */
  const int32_t* lt_type;    /* [16 T] (the same for the 16 rows of a tile) */
  const int32_t* lc_tpos;    /* [Lc]: quad-tile row of the canonical edge */
  const int32_t* lc_tmir;    /* [Lc]: ... of its mirror, or -1 */
  const int32_t* quad_wg_ptr;/* [257] or null: workgroup w of agdiff_cfconv_node's 256 persistent workgroups (k_cfconv_quad) owns the quads
                                [quad_wg_ptr[w], quad_wg_ptr[w+1]) -- contiguous ranges of like tile counts; null (or fewer workgroups):
                                equal quad counts */
  const int32_t* loc_bits;   /* [N][W], W = 2 ceil(max_atoms_per_graph / 64) 32-bit words, or null: the static local in-adjacency of
                                every atom as a bit mask over its molecule's atoms -- bit (j - graph_ptr[g]) of row i is set when
                                the local edge j -> i exists.  agdiff_sampler_front copies a molecule's rows into LDS (the radius
                                graph excludes local pairs); null: it builds them from loc_in_ptr / loc_in_eid / loc_src per step */
} agdiff_topo_t;

/* ---- workspace (device buffers the host allocates once per batch) ------------------------- */
typedef struct agdiff_ws {
  /* dynamic graph, destination-sorted: edges of target i are [in_ptr[i], in_ptr[i+1]) with src ascending */
  int32_t* num_edges;        /* [1]  E (device scalar, rewritten by every graph build) */
  int32_t* num_local;        /* [1]  L as a device scalar (written once by the host) */
  int32_t* graph_edge_cnt;   /* [G]  */
  int32_t* graph_edge_ptr;   /* [G+1] */
  int32_t* in_ptr;           /* [N+1] */
  int32_t* out_ptr;          /* [N+1]: reference order (sorted by (src,dst)): edges with src == i */
  int32_t* e_src;            /* [max_edges] */
  int32_t* e_dst;            /* [max_edges] */
  int32_t* e_type;           /* [max_edges] */
  float*   e_len;            /* [max_edges] */
  int32_t* ref2dst;          /* [max_edges]: reference position q -> destination-sorted id */
  int32_t* e_loc;            /* [max_edges]: row of l_attr_rows (= canonical index, topo->loc_row) of the edge's local (type > 0)
                                list entry, -1 for radius-only edges */
  /* canonical edges: j -> i and i -> j with equal type have the same length and type, hence bit-identical
   * edge_attr and pair-head output (dualenc.py:189-211); one of the two (src < dst) is canonical, as is every edge
   * without such a mirror.  Destination-sorted like the full list; the encoder and the global head walk this list. */
  int32_t* num_canon;        /* [1]  device scalar */
  int32_t* graph_canon_cnt;  /* [G]  */
  int32_t* graph_canon_ptr;  /* [G+1] */
  float*   c_len;            /* [max_edges] */
  int32_t* c_type;           /* [max_edges] */
  int32_t* c_src;            /* [max_edges] */
  int32_t* c_dst;            /* [max_edges] */
  int32_t* c_pos;            /* [max_edges]: the canonical edge's own id in the destination-sorted list */
  int32_t* c_mir;            /* [max_edges]: its mirror's id there, or -1 */
  float*   e_attr;           /* [ceil(max_edges/16)] tiles x 2048 floats: edge_attr in operand form (csrc/common.hpp) */
  float*   e_inv_global;     /* [max_edges] grad_global_dist_mlp output, destination-sorted */
  float*   e_scale;          /* [2*num_convs][ceil(max_edges/16)*16]: lw(d)*C(d) of conv1 / conv2 of every block (schnet.py:138-149) */
  /* local edges (reference order) */
  float*   l_len;            /* [L] */
  float*   lc_len;           /* [Lc] lengths of the canonical local edges (same values as l_len[lc_pos]) */
  int32_t* num_local_canon;  /* [1]  Lc as a device scalar (written once by the host) */
  float*   l_attr_rows;      /* fp32 row-major edge_attr rows [128] of the local edges (GIN message gather, local head): ONE row
                                per canonical local edge (row c for lc_*[c]; a mirror pair shares it: the GIN layers are
                                bound by reading these rows), written by the global encoder pass through e_loc when that
                                pass runs, else by a pass over the canonical local list.  With a caller-supplied graph
                                (AGDIFF_FWD_GRAPH_GIVEN: lengths need not be symmetric) one row per local edge [L]. */
  float*   l_inv;            /* [L] grad_local_dist_mlp output */
  /* nodes */
  float*   h;                /* [N][128] SchNet node state */
  float*   xs;               /* [N][192] lin1/BN/LeakyReLU outputs feeding conv1 (0..127) and conv2 (128..191) */
  float*   agg;              /* [N][192] CFConv aggregates */
  float*   agg_first;        /* [chunks][192], chunks = ceil(ceil(max_edges/16) / agdiff_conv_chunk_tiles(max_edges)) (agdiff_cfconv_fused):
                                partial sum of the target whose list was already open when the chunk started; the node stage
                                adds them in chunk order */
  float*   hl;               /* [N][128] GIN node state (ping) */
  float*   hl2;              /* [N][128] (pong) */
  int32_t* nan_flag;         /* [1 + G]: [0] set to 1 when any position becomes NaN, [1 + g] when one of graph g does
                                (sticky: the host clears them when a sampling job starts) */
  int32_t* range_rows;       /* [N] or null.  Split-fp16 operands saturate at 65504: the kernels that convert activations which
                                depend on the STATE and which no tensor of the workspace shows (hidden layers of the node stage, the
                                GIN layers, the pair heads) set range_rows[node] = 1 when such a value of that node (of an edge's
                                source node) reaches 65000 -- and, where they store them, when a node state (|h|, |hl| >= 255: the
                                heads multiply two) or a CFConv input (|xs| >= 60000) leaves the range the host's tensor watch
                                polls, so that an excursion between two polls is not lost.  The node's results are then not to be
                                trusted in this mode: the caller
                                polls the flags with the NaN flag, clears them and runs the owning molecules in split-bf16
                                (agdiff_amd/epsnet.py range_report).  Never written in the other modes. */
  /* CFConv by filter polynomials (agdiff_params_t.poly_kt > 0): the radius edges (type 0) of the dynamic graph by TARGET, in
   * AGDIFF_RAD_STRIDE rows per target (written by agdiff_graph_build next to the full list; sources ascending).  Rows
   * [rad_cnt[i], 16 ceil(rad_cnt[i] / 16)) of target i are pad rows (src = i, length 0, scale 0: agdiff_edge_scales_split
   * writes them), rows beyond are never read by k_cfconv_node: every 16-row tile belongs to ONE target.  On quads
   * (topo->group_targets == 4, p->tune_cfconv_quad_tiles >= 0: k_cfconv_quad) agdiff_sampler_front does NOT write the pad rows:
   * that kernel runs rows [rad_cnt[i], 4 ceil(max over the quad / 4)) with scale 0 itself and only needs them to hold valid
   * indices and finite numbers -- older rows of the same molecule, or the zeros the host allocated the buffers with. */
  int32_t* rad_cnt;          /* [N]  radius edges of target i (<= AGDIFF_RADIUS_CAP) */
  int32_t* rad_src;          /* [N * AGDIFF_RAD_STRIDE] */
  float*   rad_len;          /* [N * AGDIFF_RAD_STRIDE] */
  float*   r_scale;          /* [2*num_convs][N * AGDIFF_RAD_STRIDE]: lw(d)*C(d) by radius-list row */
  int32_t* num_local_padded; /* [1]  Lp as a device scalar (written once by the host) */
  float*   l_scale;          /* [2*num_convs][ceil(Lp/16)*16]: the same by padded-list position (pad entries stay 0) */
  float*   l_attr_frag;      /* [ceil(Lp/16)] tiles x 2048 floats: edge_attr of the local edges in operand form, by padded-list
                                position (only written / read when the local edges go through the filter MLPs) */
  float*   l_len_p;          /* [Lp] lengths of the local edges by padded-list position (agdiff_local_lengths; pads stay 0) */
  int32_t* g_inbits;         /* [N][2 * ceil(max_atoms_per_graph / 64)] hand-over from the graph build's count pass to its fill
                                pass: row i = in-adjacency bit mask of atom i inside its molecule (optional, with g_deg / g_cdeg) */
  int32_t* g_deg;            /* [N] in-degrees */
  int32_t* g_cdeg;           /* [N] canonical in-degrees */
  int32_t* enc_flags;        /* [1 + ceil(Lc/16)]: agdiff_local_edge_rows: [0] = tiles of the canonical local list that hold an edge
                                longer than the cutoff (or of a type without a polynomial) this step, [1 + tile] = the 16-bit mask
                                of those rows: the encoder MLP evaluates exactly them, the polynomials all the others */
  float*   h0;               /* [N][128] cache of node stage 0's h (the atom embeddings: they do not depend on the positions) */
  float*   xs0;              /* [N][192] ... and of its xs (block 0's lin1 / BN / LeakyReLU outputs) */
  float*   agg_loc;          /* [N][192] CFConv aggregates over the local edges (agdiff_cfconv_local) */
  float*   agg_first_loc;    /* [ceil(ceil(Lp/16) / agdiff_conv_chunk_tiles(Lp))][192] */
  float*   lt_len;           /* [16 T] lengths of the local edges by quad-tile row (agdiff_local_lengths; pads stay 0) */
  float*   lt_scale;         /* [2*num_convs][16 T]: lw(d)*C(d) by quad-tile row (pad rows stay 0) */
  float*   inv_r;            /* [N * AGDIFF_RAD_STRIDE] grad_global_dist_mlp output by radius row (fused sampler front) */
  int32_t* canon_counter;    /* [2] live length of the fused sampler front's canonical radius list, by step parity: every molecule
                                claims its range of ws->c_* with one atomic add on [parity] (and molecule 0 zeroes the other) */
  int64_t* variant_log;      /* [host] one word or null: every launcher ORs the AGDIFF_VAR_* bit of the variant it chose */
} agdiff_ws_t;

typedef struct agdiff_step_args {
  const float* pos_in;       /* [N][3] */
  float*       pos_out;      /* [N][3] (may alias pos_in) */
  float*       scratch;      /* [N][3] uncentred positions between the two passes of the update */
  const float* noise;        /* [N][3] standard normal draws for this step (torch.randn_like, dualenc.py:529) */
  float*       traj_out;     /* [N][3] or null: copy of the centred positions (pos_traj, dualenc.py:545) */
  float sigma;               /* sigmas[i] */
  float step_size;           /* step_lr * (sigmas[i] / 0.01) ** 2, evaluated by the host in fp32 (dualenc.py:532) */
  float noise_scale;         /* sqrt(step_size * 2) (dualenc.py:536) */
  float w_global;
  float clip;                /* clip_norm limit of the global term (dualenc.py:522) */
  float clip_local;          /* < 0: no local clipping (clip_local=None) */
  float clip_pos;            /* < 0: no clamp */
  int32_t use_global;        /* sigmas[i] < global_start_sigma (dualenc.py:515) */
} agdiff_step_args_t;

/* Tiles per chunk the fused CFConv kernel and the node stage use for a workspace of `max_edges` edges (1..8):
 * small batches get short chunks so that every wave slot of the chip has work. */
int agdiff_conv_chunk_tiles(int64_t max_edges);

/* Host helper of the topology builder (no GPU work): the order of a molecule's n atoms in which consecutive runs of `gt`
 * atoms form the target groups agdiff_cfconv_node's waves own (agdiff_topo_t.quad_tgt).  need [n][k] (row-major): local
 * 16-row tiles atom i needs of local edge type k = ceil(in-edges of that type / (16 / gt)); a group costs the sum over the
 * types of the MAX of its atoms' needs, so atoms with like needs belong together.  Search: the atoms sorted by their need
 * vectors with each type in turn as the leading key, or by total need; from every start, swaps of two atoms between two
 * groups while the total falls; the cheapest result wins (ties: the earliest start).  Deterministic.  order_out [n]. */
int agdiff_group_order(const int32_t* need /* [host] */, int32_t n, int32_t k, int32_t gt, int32_t* order_out /* [host] */);

/* Build stamp / ABI check. */
int agdiff_abi_version(void);
/* sizeof() of every struct above, for the binding's self-check: out[0..6] =
 * conv, gin, head, params, topo, ws, step_args. */
int agdiff_struct_sizes(int64_t* out /* [host] */);

/* torch_cluster.radius_graph + sparse add/coalesce (models/common.py:208-233) + get_distance
 * (geometry.py:5-6): rebuilds the destination-sorted edge list, its reference-order permutation,
 * edge types and lengths from `pos`. */
int agdiff_graph_build(const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, float cutoff, void* stream);
/* The same with canon_radius_only != 0: the canonical list (ws->c_*, num_canon) holds RADIUS edges only -- one of j -> i /
 * i -> j when both are radius edges.  What the denoising loop asks for (AGDIFF_FWD_SAMPLER with poly_kt > 0): there only
 * the global head walks the canonical list, and only radius edges' results are used. */
int agdiff_graph_build_ex(const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, float cutoff,
                          int32_t canon_radius_only, void* stream);

/* The same with the radius rows' CFConv scales (ws->r_scale, all 2 * num_convs of them: DistanceWeightingNetwork x cutoff
 * envelope, encoder/schnet.py:83-100,138-149) and the pad rows of every target's last tile written by the fill pass itself:
 * what the denoising loop calls (one launch less per step than agdiff_graph_build_ex + agdiff_edge_scales_split(0)). */
int agdiff_graph_build_scaled(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos,
                              float cutoff, int32_t canon_radius_only, void* stream);

/* get_distance on the static local edges (geometry.py:5-6 applied to edge_index[:, local_edge_mask]): one evaluation per
 * canonical local edge, written to l_len of the edge and of its mirror, to lc_len and -- where the workspace has them --
 * to l_len_p (padded-list positions) and lt_len (quad-tile rows). */
int agdiff_local_lengths(const agdiff_topo_t* topo, const agdiff_ws_t* ws, const float* pos, void* stream);

/* DistanceWeightingNetwork x cutoff envelope of all 2*num_convs CFConvs (encoder/schnet.py:83-100, 138-149):
 * ws->e_scale from ws->e_len; per_canonical_edge != 0: from ws->c_len, one evaluation per canonical edge written to its
 * position and its mirror's (equal lengths: the same value bit for bit). */
int agdiff_edge_scales(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                       int32_t per_canonical_edge, void* stream);

/* get_edge_encoder(cfg)(edge_length, edge_type) (encoder/edge.py:106-116): MLPEdgeEncoder.forward
 * (edge.py:84-103) when p->edge_encoder == 0, GaussianSmearingEdgeEncoder.forward (edge.py:34-42) when 1.
 * n_edges_dev: device scalar with the live edge count (<= max_tiles*16).  Outputs, each optional:
 *   attr_frag  operand-form edge_attr tiles;
 *   attr_rows  fp32 rows [.][128]: row e of edge e, or row row_index[e] when row_index is given (negative: no
 *              row) -- how the pass over all edges also serves the local edge list.
 * pos_index / mir_index (both or neither): the n edges are a canonical list (agdiff_ws_t.c_*); edge e's results go
 * to position pos_index[e] and, when >= 0, mir_index[e] of attr_frag / row_index instead of position e. */
int agdiff_edge_encoder(const agdiff_params_t* p, const int32_t* n_edges_dev, int64_t max_tiles,
                        const float* e_len, const int32_t* e_type, float* attr_frag, float* attr_rows,
                        const int32_t* row_index, const int32_t* pos_index, const int32_t* mir_index, void* stream);

/* edge_attr rows of the canonical local edges (ws->l_attr_rows, one row per canonical edge: what the GIN layers and the
 * local head read; dualenc.py:214-216): with per-type polynomials available (p->attr_poly_typed_pk, p->poly_num_slots > 0)
 * every 16-edge tile whose lengths all lie inside [0, cutoff] takes its rows from them; a tile with a longer edge (bonded
 * atoms far apart at high sigma) is flagged and goes through agdiff_edge_encoder's MLP.  Without polynomials: the MLP for
 * all of them.  Needs ws->lc_len (agdiff_local_lengths). */
int agdiff_local_edge_rows(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, void* stream);

/* Node-side stage k of SchNetEncoder.forward (encoder/schnet.py:268-282): k == 0 embeds atoms;
 * k >= 1 finishes InteractionBlock k-1 (lin2/BN, act, lin, gate, AdaptiveScaling, residual);
 * k < num_convs also applies block k's conv{1,2}.lin1/BN/LeakyReLU into ws->xs.  Block k-1's aggregates are ws->agg /
 * ws->agg_first by chunks of the full edge list (agdiff_cfconv_fused). */
int agdiff_schnet_node_stage(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t k, void* stream);
/* The same with `split` bits: 1: block k-1's aggregates are ws->agg as agdiff_cfconv_node wrote it (one complete row per
 * node); 8 (with 1): plus ws->agg_loc / ws->agg_first_loc of agdiff_cfconv_local (lists topo->lp_ptr).  4 (k == 0): write
 * the stage's h / xs to the cache ws->h0 / ws->xs0; 2 (k == 1): block 0's input h is ws->h0. */
int agdiff_schnet_node_stage_split(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t k,
                                   int32_t split, void* stream);

/* CFConv filter generation + message + aggr='add' for both convs of block k
 * (encoder/schnet.py:138-162; PyG MessagePassing.propagate): ws->agg / ws->agg_first. */
int agdiff_cfconv_fused(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t k, void* stream);

/* The same two CFConvs with the filters from d-polynomials (agdiff_params_t.poly_kt > 0), one launch, one wave per QUAD of
 * targets: the radius rows of each (ws->rad_*, ws->r_scale; p->conv[k].filt_poly_pk) and -- when agdiff_local_poly_enabled
 * -- the quad's local tiles (topo->lt_*, ws->lt_len, ws->lt_scale; per-type sets p->conv[k].filt_poly_typed_pk) are
 * summed in registers and ws->agg[i] is written once, complete, for every node (zeros for a node without edges).  Without
 * local polynomials the local edges' part comes from agdiff_cfconv_local: the filter MLPs over the padded local list
 * (topo->lp_*, ws->l_scale, ws->l_attr_frag) -> ws->agg_loc / ws->agg_first_loc, added by the node stage (split bit 8).
 * agdiff_edge_scales_split fills ws->r_scale and the radius pad rows (which == 0), ws->l_scale (1) or ws->lt_scale (2). */
int agdiff_cfconv_node(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t k, void* stream);
int agdiff_cfconv_local(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t k, void* stream);
/* How the local edges' CFConv filters are evaluated for this (model, batch):
 *   1  all by per-type d-polynomials inside agdiff_cfconv_node (every local type of the batch has a slot; topo->lt_*,
 *      ws->lt_len, ws->lt_scale present; not switched off by p->tune_local_poly_off);
 *   2  mixed: the slotted types as in 1, the others by agdiff_cfconv_local (filter MLPs on ws->l_attr_frag, where
 *      agdiff_edge_scales_split(which = 1) leaves the slotted types' scales at 0);
 *   0  all by agdiff_cfconv_local. */
int agdiff_local_poly_enabled(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws);
int agdiff_edge_scales_split(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t which,
                             void* stream);

/* agdiff_pair_head with p->head_global for edges whose edge_attr is MLPEdgeEncoder(len, type 0): the edge_attr half of the
 * first layer comes from the d-polynomial head_global.attr_poly_pk (needs p->poly_kt > 0).  pos_index / mir_index as in
 * agdiff_pair_head (both or neither). */
int agdiff_pair_head_poly(const agdiff_params_t* p, const int32_t* n_edges_dev, int64_t max_tiles,
                          const int32_t* src, const int32_t* dst, const float* len, const float* node_h,
                          const int32_t* pos_index, const int32_t* mir_index, float* out, void* stream);

/* assemble_atom_pair_feature + grad_*_dist_mlp (models/common.py:106-109, 86-103; dualenc.py:203-211,
 * 226-239) over n edges given by (src, dst); edge_attr either as operand-form tiles (attr_frag) or as fp32
 * rows [n][128] (attr_rows) -- exactly one of the two.  pos_index / mir_index (both or neither, with attr_frag):
 * the n edges are a canonical list; edge e reads its attrs at position pos_index[e] and writes its result to
 * out[pos_index[e]] and, when >= 0, out[mir_index[e]] (h_i * h_j is symmetric, so the mirror's value is the same).
 * With attr_rows the edge's attributes are row e of attr_rows, whatever pos_index is (the local head over the canonical
 * local list: row c, results to out[lc_pos[c]] and out[lc_mir[c]]). */
int agdiff_pair_head(const agdiff_head_params_t* hp, const int32_t* n_edges_dev, int64_t max_tiles,
                     const int32_t* src, const int32_t* dst, const float* node_h, const float* attr_frag,
                     const float* attr_rows, const int32_t* pos_index, const int32_t* mir_index, float* out,
                     void* stream);

/* GINEncoder.forward (encoder/gin.py:112-148) on the static local edges; result in ws->hl.
 * rows_per_canonical_edge != 0: ws->l_attr_rows holds one row per canonical local edge (local edge e reads row
 * topo->loc_row[e]); 0: one row per local edge. */
int agdiff_gin_encoder(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                       int32_t rows_per_canonical_edge, void* stream);

/* PyG MessagePassing.propagate(aggr='add') with message x_j * W (encoder/schnet.py:156,161-162) as a
 * stand-alone op on a destination-sorted CSR: out[i][:] = sum_{e in [in_ptr[i], in_ptr[i+1])} x[src[e]][:] * W[e][:].
 * F must be 64 or 128. */
int agdiff_cfconv_aggregate(const float* x, const float* W, const int32_t* in_ptr, const int32_t* src,
                            int64_t num_nodes, int32_t F, float* out, void* stream);

/* The whole score network, dualenc.py:142-251.  flags:
 *   AGDIFF_FWD_GLOBAL       run the global branch; without it everything whose result the sampler discards
 *                           when sigma >= global_start_sigma (dualenc.py:523-524) is skipped
 *   AGDIFF_FWD_NO_RADIUS    extend_radius=False (dualenc.py:166-176): the graph is the bond graph only
 *   AGDIFF_FWD_GRAPH_GIVEN  caller-supplied edge_index / edge_type / edge_length (dualenc.py:165): ws already
 *                           holds the destination-sorted graph (num_edges, in_ptr, e_src/e_dst/e_type/e_len)
 *                           and l_len; neither is rebuilt from `pos` */
#define AGDIFF_FWD_GLOBAL 1
#define AGDIFF_FWD_NO_RADIUS 2
#define AGDIFF_FWD_GRAPH_GIVEN 4
/*   AGDIFF_FWD_SAMPLER      the caller is the denoising loop: of the global head's outputs only those of radius edges are
 *                           used (edge_inv_global * (1 - local_edge_mask), dualenc.py:516-518), so with
 *                           poly_kt > 0 neither the edge encoder nor the head runs on anything but d-polynomials and the
 *                           local list; ws->e_inv_global is then only valid at radius edges */
#define AGDIFF_FWD_SAMPLER 8
/*   AGDIFF_FWD_STAGE0_CACHED ws->h0 / ws->xs0 already hold node stage 0's outputs for this topology and these weights (an
 *                           earlier call with AGDIFF_FWD_GLOBAL on the same workspace wrote them): stage 0 -- embedding
 *                           look-up and block 0's lin1, which do not depend on `pos` -- is not launched again */
#define AGDIFF_FWD_STAGE0_CACHED 16
/*   AGDIFF_FWD_GRAPH_READY  (with AGDIFF_FWD_SAMPLER, poly_kt > 0) agdiff_sampler_front has already built this step's radius rows,
 *                           their scales and its canonical radius list from `pos` (and written the local edges' lengths and
 *                           quad-tile scales): no graph build here; the global head's outputs go to ws->inv_r (by radius
 *                           row), where the next agdiff_sampler_front reads them.  AGDIFF_FWD_PARITY: the step's parity bit
 *                           (which of ws->canon_counter[2] holds the list's length) */
#define AGDIFF_FWD_GRAPH_READY 32
#define AGDIFF_FWD_PARITY 64
/*   AGDIFF_FWD_GRAPH_PENDING (with AGDIFF_FWD_GRAPH_READY) the front has run its update and local phases only (mode 1 | 4): the
 *                           graph phase (mode 2) is launched here, on `stream`, AFTER the local branch has been forked onto the
 *                           side stream -- the local branch then runs beside it instead of beside the first CFConv */
#define AGDIFF_FWD_GRAPH_PENDING 128
int agdiff_score_forward(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                         const float* pos, int32_t flags, void* stream);

/* The serial front of a denoising step in ONE launch (csrc/front.hip), one workgroup per molecule:
 *   mode & 1  the Langevin update of the step described by `s` (eq_transform x 2, clip_norm, move, NaN check, center_pos,
 *             clamp, trajectory row: geometry.py:9-17, dualenc.py:506-545, 581-589) from ws->l_inv and -- when
 *             s->use_global -- ws->inv_r over the radius rows the last graph phase wrote;
 *   mode & 2  then the radius graph of the NEXT forward on the positions just written (s->pos_out; s->pos_in without an
 *             update): ws->rad_cnt / rad_src / rad_len / r_scale with pad rows, and the canonical radius list ws->c_len /
 *             c_src / c_dst / c_pos / c_mir (c_pos / c_mir = radius rows) of live length ws->canon_counter[parity], parity =
 *             (mode >> 4) & 1 alternating from step to step.  `cutoff` = p->cutoff, or 0 for extend_radius = False;
 *   mode & 4  and, on the same positions, what agdiff_local_lengths and agdiff_edge_scales_split(which = 2) write: the local
 *             edges' lengths in every layout (ws->l_len, lc_len, l_len_p, lt_len) and their CFConv scales by quad-tile row
 *             (ws->lt_scale) -- agdiff_score_forward then skips both (AGDIFF_FWD_GRAPH_READY).
 * The following agdiff_score_forward takes AGDIFF_FWD_SAMPLER | AGDIFF_FWD_GRAPH_READY.  Replaces agdiff_langevin_update +
 * agdiff_graph_build_scaled inside the denoising loop (models/common.py:208-233 is rebuilt every step). */
int agdiff_sampler_front(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                         const agdiff_step_args_t* s, int32_t mode, float cutoff, void* stream);
/* ONE denoising step of the loop above as a replayable HIP graph -- [agdiff_sampler_front(front_mode) | agdiff_score_forward(fwd_flags
 * on host_step->pos_in) | step counter + 1], captured from `stream` (side stream of the local branch included) and instantiated:
 * small batches (scripts/test.py:130-164 samples one molecule's 100..1000 conformers per call) are bound by the ~25 launches a step
 * takes, a replay is one call.  What changes from step to step does not sit in kernel arguments: the front kernel reads the
 * update's step from the DEVICE table `step_table` at index *step_index (the host fills the table for the whole run: noise and
 * trajectory rows, sigma, step size, ...), and the last node of the graph increments *step_index.  `host_step` only carries what
 * is the same for every step (pos_in / pos_out, and valid pointers for the argument checks).  One graph per (front_mode, fwd_flags)
 * the loop alternates between (step parity, global branch on / off).  The kernels must have run once outside a capture (their
 * function attributes, the side stream and its events are set up on first use).  *graph_out: opaque handle for
 * agdiff_step_graph_launch / agdiff_step_graph_destroy. */
int agdiff_step_graph_capture(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                              const agdiff_step_args_t* host_step /* [host] */, const agdiff_step_args_t* step_table,
                              int32_t* step_index, int32_t front_mode, float cutoff, int32_t fwd_flags, void* stream,
                              void** graph_out /* [host] */);
int agdiff_step_graph_launch(void* graph /* [host] */, void* stream);
int agdiff_step_graph_destroy(void* graph /* [host] */);

/* The polynomial global head (agdiff_pair_head_poly) over the canonical radius list of agdiff_sampler_front (step parity as
 * there): results to ws->inv_r at the entry's radius row and its mirror's. */
int agdiff_pair_head_poly_rows(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws, int32_t parity,
                               void* stream);

/* In-step timing of the dominant kernel (bench.py's roofline object): between agdiff_profile_cfconv(1) and (0) every CFConv
 * launch that agdiff_score_forward issues for the global branch (one InteractionBlock's agdiff_cfconv_node [+ agdiff_cfconv_local],
 * or agdiff_cfconv_fused) is bracketed by a pair of HIP events on the stream it runs on; agdiff_profile_cfconv_read waits for
 * them, returns their summed elapsed time and the number of bracketed launches, and resets the list.  [host] pointers. */
int agdiff_profile_cfconv(int32_t enable);
int agdiff_profile_cfconv_read(double* total_ms, int64_t* launches);

/* eq_transform x2, clip_norm, Langevin update, NaN check, center_pos, clamp
 * (geometry.py:9-17; dualenc.py:506-545, 581-589) from ws->l_inv / ws->e_inv_global. */
int agdiff_langevin_update(const agdiff_topo_t* topo, const agdiff_ws_t* ws, const agdiff_step_args_t* a, void* stream);

/* pos_perturbed = pos + pos_noise * sqrt(1 - a) / sqrt(a) with a = alpha_graph[graph of the atom]
 * (get_loss_diffusion, dualenc.py:306-312; alpha_graph[G] = alphas.index_select(0, time_step)). */
int agdiff_perturb_positions(const agdiff_topo_t* topo, const float* pos, const float* noise,
                             const float* alpha_graph, float* pos_out, void* stream);

/* Forward value of get_loss_diffusion (dualenc.py:329-395; evaluated under no_grad by scripts/train.py:160-170)
 * after agdiff_score_forward(pos_perturbed, AGDIFF_FWD_GLOBAL): d_gt / d_target per edge, global_mask, the four
 * eq_transforms and the per-atom squared errors.  loss: [3][N] = total, 2*global, 5*local. */
int agdiff_diffusion_loss(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                          const float* pos_gt, const float* pos_perturbed, const float* alpha_graph,
                          float* loss, void* stream);

/* One denoising step = agdiff_score_forward + agdiff_langevin_update (dualenc.py:478-545). */
int agdiff_langevin_step(const agdiff_params_t* p, const agdiff_topo_t* topo, const agdiff_ws_t* ws,
                         const agdiff_step_args_t* a, void* stream);

/* ---- evaluation (the step after the path) ------------------------------------------------------
 * get_rmsd_confusion_matrix (utils/evaluation/covmat.py:16-35): out[j][i] = GetBestRMS(gen_i, ref_j) over the atoms
 * listed in `atom_idx` (the reference removes hydrogens first, utils/chem.py:133-137) = the smallest, over the atom
 * mappings `perms`, of the RMSD after the optimal proper rotation + translation (rdkit rdMolAlign.GetBestRMS: every
 * substructure match of the molecule onto itself is aligned with AlignMol, no reflection, uniform weights).
 *   pos_ref [R][n][3], pos_gen [G][n][3]   conformers of ONE molecule with n atoms
 *   atom_idx [m]                            atoms that take part (m <= AGDIFF_RMSD_MAX_ATOMS), e.g. the heavy atoms
 *   perms [P][m] or null                    mapping p pairs generated atom atom_idx[k] with reference atom
 *                                           atom_idx[perms[p][k]]; null = the identity only (no symmetry: an upper bound
 *                                           of GetBestRMS)
 *   scratch [(R + G) * (3 m + 1)] floats    centred coordinates of the selected atoms (written by the call)
 *   out [R][G] */
int agdiff_rmsd_matrix(const float* pos_ref, const float* pos_gen, const int32_t* atom_idx, const int32_t* perms,
                       int32_t R, int32_t G, int32_t n, int32_t m, int32_t P, float* scratch, float* out, void* stream);

/* Row and column minima of a confusion matrix [R][G] (covmat.py:135-136: rmsd_ref_min = min over generated,
 * rmsd_gen_min = min over references): row_min [R], col_min [G]. */
int agdiff_matrix_minima(const float* mat, int32_t R, int32_t G, float* row_min, float* col_min, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AGDIFF_HIP_H */
