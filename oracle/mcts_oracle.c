/*
 * mcts_oracle.c -- plain-C CPU oracle of the MCTS hot path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from this
 * file (oracle/_build/libmzoracle.so); muzero_amd/ never links or calls it.
 *
 * It restates the reference algorithm (fidel-schaposnik/muzero) node by node, with explicit parent links
 * like the reference's Node objects, one tree at a time.  Parity is PINNED: tests/test_oracle_golden.py
 * replays every fixture of tests/golden/ (generated from the unmodified reference agent) through it.
 *
 * Reference lines restated (relative to /root/reference):
 *   Node.update_value                 agent.py:119-123
 *   MinMaxStats.update / normalize    utils.py:63-76
 *   exploration_function              config.py:159-160
 *   effective discount                agent.py:150-152
 *   expand_node + masked_policy       agent.py:154-160, network.py:142-146
 *   add_exploration_noise             agent.py:170-175
 *   select_leaf/select_child/ucb      agent.py:177-190
 *   backpropagate                     agent.py:192-196
 *   softmax_sample / select_action    agent.py:162-168,213-217
 *   policy target / root value        agent.py:229-231
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math (oracle/Makefile).  x86-64 SSE2 doubles are
 * IEEE; contraction is off so reward + discount*value rounds twice, as in CPython.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t parent;       /* node index, -1 for the root */
    int32_t first_child;  /* node index of child 0, -1 while not expanded */
    int32_t num_children;
    int32_t action;       /* action leading here */
    int32_t visit_count;
    int32_t block;        /* expansion index (network call that expanded it), -1 while not expanded */
    double value_sum;
    double prior;         /* fp32 widened; fp64 at root children after the noise mix */
    double reward;        /* fp32 widened; NaN until expanded (reference: None) */
} node_t;

typedef struct {
    node_t* nodes;
    int32_t num_nodes;
    int32_t leaf;         /* node selected by the last select */
    int32_t blocks;       /* expansions so far (root = block 0) */
    double mm_min, mm_max;
    int64_t sum_path_len, sims;
} tree_t;

typedef struct mzo_pool {
    int32_t T, A, N, has_bounds;
    double discount, default_value, kmin, kmax;
    double* log_table;
    tree_t* trees;
    int32_t blocks_in_use;
} mzo_pool_t;

typedef struct {
    int32_t num_trees, num_actions, max_simulations, num_players, has_known_bounds, reserved;
    double discount, default_value, known_min, known_max;
    const double* log_table;
} mzo_config_t;

/* NumPy float32 pairwise summation (numpy/_core/src/umath/loops_utils.h.src), trap T4. */
static float pairwise_sum_f32(const float* a, int64_t n) {
    if (n < 8) {
        float res = -0.0f;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        int64_t i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
    }
}

float mzo_numpy_sum_f32(const float* a, int64_t n) { return pairwise_sum_f32(a, n); }

/* network.py:142-146 given exp(logits) already evaluated (the exp itself is not reproducible, trap T4). */
void mzo_masked_policy_from_exp(const float* exp_logits, const int32_t* actions, int32_t n_actions, int32_t A,
                                float* out_policy) {
    float* masked = (float*)calloc((size_t)A, sizeof(float));
    for (int32_t i = 0; i < n_actions; ++i) masked[actions[i]] = exp_logits[actions[i]];
    const float total = pairwise_sum_f32(masked, A);
    for (int32_t a = 0; a < A; ++a) out_policy[a] = masked[a] / total;
    free(masked);
}

int mzo_pool_create(const mzo_config_t* cfg, mzo_pool_t** out) {
    mzo_pool_t* p = (mzo_pool_t*)calloc(1, sizeof(mzo_pool_t));
    if (!p) return -2;
    p->T = cfg->num_trees;
    p->A = cfg->num_actions;
    p->N = cfg->max_simulations;
    p->has_bounds = cfg->has_known_bounds;
    p->discount = cfg->num_players == 2 ? -cfg->discount : cfg->discount; /* agent.py:150-152 */
    p->default_value = cfg->default_value;
    p->kmin = cfg->known_min;
    p->kmax = cfg->known_max;
    p->log_table = (double*)malloc(sizeof(double) * (size_t)(p->N + 2));
    for (int n = 0; n < p->N + 2; ++n)
        p->log_table[n] = cfg->log_table ? cfg->log_table[n] : 2 * log((double)(n + 1)); /* config.py:160 */
    p->trees = (tree_t*)calloc((size_t)p->T, sizeof(tree_t));
    const size_t max_nodes = 1 + (size_t)(p->N + 1) * (size_t)p->A;
    for (int t = 0; t < p->T; ++t) {
        p->trees[t].nodes = (node_t*)malloc(sizeof(node_t) * max_nodes);
        if (!p->trees[t].nodes) return -2;
    }
    *out = p;
    return 0;
}

void mzo_pool_destroy(mzo_pool_t* p) {
    if (!p) return;
    for (int t = 0; t < p->T; ++t) free(p->trees[t].nodes);
    free(p->trees);
    free(p->log_table);
    free(p);
}

static void init_node(node_t* n, int32_t parent, int32_t action, double prior) {
    n->parent = parent;
    n->first_child = -1;
    n->num_children = 0;
    n->action = action;
    n->visit_count = 0;
    n->block = -1;
    n->value_sum = 0.0;
    n->prior = prior;
    n->reward = NAN;
}

/* expand_node on the root (agent.py:154-160): child k <- (legal[k], priors[k]) positionally (trap T5). */
int mzo_expand_roots(mzo_pool_t* p, const float* priors, const int32_t* legal, const int32_t* n_legal) {
    for (int t = 0; t < p->T; ++t) {
        tree_t* tr = &p->trees[t];
        const int n = n_legal ? n_legal[t] : p->A;
        init_node(&tr->nodes[0], -1, -1, 1.0);
        tr->num_nodes = 1;
        tr->leaf = -1;
        tr->blocks = 0;
        tr->mm_min = INFINITY;   /* utils.py:60-61 */
        tr->mm_max = -INFINITY;
        if (n <= 0) continue;    /* inactive tree */
        tr->nodes[0].first_child = 1;
        tr->nodes[0].num_children = n;
        tr->nodes[0].block = 0;
        for (int k = 0; k < n; ++k)
            init_node(&tr->nodes[1 + k], 0, legal ? legal[(size_t)t * p->A + k] : k, (double)priors[(size_t)t * p->A + k]);
        tr->num_nodes = 1 + n;
        tr->blocks = 1;
    }
    p->blocks_in_use = 1;
    return 0;
}

/* agent.py:170-175 */
int mzo_add_root_noise(mzo_pool_t* p, const double* noise, double frac) {
    for (int t = 0; t < p->T; ++t) {
        tree_t* tr = &p->trees[t];
        for (int k = 0; k < tr->nodes[0].num_children; ++k) {
            node_t* c = &tr->nodes[1 + k];
            c->prior = c->prior * (1 - frac) + noise[(size_t)t * p->A + k] * frac;
        }
    }
    return 0;
}

static double normalize(const mzo_pool_t* p, const tree_t* tr, double value) { /* utils.py:68-76 */
    double a, b;
    if (!p->has_bounds) {
        if (tr->mm_max == tr->mm_min) return value;
        a = tr->mm_min;
        b = tr->mm_max;
    } else {
        a = p->kmin;
        b = p->kmax;
    }
    return (value - a) / (b - a);
}

static double ucb_score(const mzo_pool_t* p, const tree_t* tr, const node_t* node) { /* agent.py:187-190 */
    const node_t* parent = &tr->nodes[node->parent];
    const double value = node->visit_count > 0 ? node->value_sum / node->visit_count : NAN;
    const double exploit = isnan(value) ? p->default_value : node->reward + p->discount * value;
    const double explore = node->prior * (p->log_table[parent->visit_count] / (node->visit_count + 1));
    return normalize(p, tr, exploit) + explore;
}

/* select_leaf (agent.py:177-185): Python max over (score, action, child) tuples. */
int mzo_select(mzo_pool_t* p, int32_t* out_parent_block, int32_t* out_action) {
    for (int t = 0; t < p->T; ++t) {
        tree_t* tr = &p->trees[t];
        if (tr->blocks == 0) {
            out_parent_block[t] = -1;
            out_action[t] = -1;
            continue;
        }
        int32_t node = 0, depth = 0;
        while (tr->nodes[node].num_children > 0) {
            const node_t* nd = &tr->nodes[node];
            int32_t best = nd->first_child;
            double best_score = ucb_score(p, tr, &tr->nodes[best]);
            for (int k = 1; k < nd->num_children; ++k) {
                const int32_t c = nd->first_child + k;
                const double s = ucb_score(p, tr, &tr->nodes[c]);
                if (s > best_score || (s == best_score && tr->nodes[c].action > tr->nodes[best].action)) {
                    best = c;
                    best_score = s;
                }
            }
            node = best;
            ++depth;
        }
        tr->leaf = node;
        tr->sum_path_len += depth;
        out_parent_block[t] = tr->nodes[tr->nodes[node].parent].block;
        out_action[t] = tr->nodes[node].action;
    }
    return 0;
}

static void mm_update(const mzo_pool_t* p, tree_t* tr, double value) { /* utils.py:63-66 */
    if (p->has_bounds) return;
    if (value > tr->mm_max) tr->mm_max = value;
    if (value < tr->mm_min) tr->mm_min = value;
}

/* expand_node(leaf, all actions) + backpropagate (agent.py:208-211,154-160,192-196). */
int mzo_expand_backprop(mzo_pool_t* p, const float* priors, const float* reward, const float* value) {
    for (int t = 0; t < p->T; ++t) {
        tree_t* tr = &p->trees[t];
        if (tr->blocks == 0) continue;
        node_t* leaf = &tr->nodes[tr->leaf];
        leaf->reward = (double)reward[t];
        leaf->first_child = tr->num_nodes;
        leaf->num_children = p->A;
        leaf->block = tr->blocks++;
        for (int k = 0; k < p->A; ++k)
            init_node(&tr->nodes[tr->num_nodes + k], tr->leaf, k, (double)priors[(size_t)t * p->A + k]);
        tr->num_nodes += p->A;
        double v = (double)value[t];
        int32_t node = tr->leaf;
        while (node >= 0) {
            node_t* nd = &tr->nodes[node];
            nd->value_sum += v;
            nd->visit_count += 1;
            mm_update(p, tr, nd->value_sum / nd->visit_count);
            v = nd->parent >= 0 ? nd->reward + p->discount * v : NAN;   /* root: reward None */
            node = nd->parent;
        }
        tr->sims += 1;
    }
    p->blocks_in_use += 1;
    return 0;
}

int mzo_root_stats(mzo_pool_t* p, int32_t* visits, double* root_value, double* minmax, float* policy) {
    for (int t = 0; t < p->T; ++t) {
        const tree_t* tr = &p->trees[t];
        const node_t* root = &tr->nodes[0];
        for (int a = 0; a < p->A; ++a) {
            if (visits) visits[(size_t)t * p->A + a] = 0;
            if (policy) policy[(size_t)t * p->A + a] = 0.f;
        }
        for (int k = 0; k < root->num_children; ++k) {
            const node_t* c = &tr->nodes[1 + k];
            if (visits) visits[(size_t)t * p->A + c->action] = c->visit_count;
            if (policy) policy[(size_t)t * p->A + c->action] = (float)((double)c->visit_count / (double)root->visit_count);
        }
        if (root_value) root_value[t] = root->visit_count > 0 ? root->value_sum / root->visit_count : NAN;
        if (minmax) {
            minmax[2 * t] = tr->mm_min;
            minmax[2 * t + 1] = tr->mm_max;
        }
    }
    return 0;
}

/* softmax_sample (agent.py:162-168) + CPython random.choices: bisect_right(accumulate(w), u*total, 0, n-1). */
int mzo_select_action(mzo_pool_t* p, const double* u, const double* temperature, int32_t* out_action) {
    double* cum = (double*)malloc(sizeof(double) * (size_t)p->A);
    for (int t = 0; t < p->T; ++t) {
        const tree_t* tr = &p->trees[t];
        const int n = tr->nodes[0].num_children;
        if (n <= 0) {
            out_action[t] = -1;
            continue;
        }
        const node_t* ch = &tr->nodes[1];
        if (temperature[t] == 0.0) {
            int best = 0;
            for (int k = 1; k < n; ++k)
                if (ch[k].visit_count > ch[best].visit_count ||
                    (ch[k].visit_count == ch[best].visit_count && ch[k].action > ch[best].action))
                    best = k;
            out_action[t] = ch[best].action;
            continue;
        }
        const double inv_t = 1 / temperature[t];
        for (int k = 0; k < n; ++k) {
            const double w = pow((double)ch[k].visit_count, inv_t);
            cum[k] = k == 0 ? w : cum[k - 1] + w;
        }
        const double x = u[t] * (cum[n - 1] + 0.0);
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (x < cum[mid]) hi = mid;
            else lo = mid + 1;
        }
        out_action[t] = ch[lo].action;
    }
    free(cum);
    return 0;
}

int mzo_pool_counters(mzo_pool_t* p, int64_t* out4) {
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    for (int t = 0; t < p->T; ++t) {
        out4[0] += p->trees[t].sims;
        out4[1] += p->trees[t].sum_path_len;
    }
    return 0;
}

/* Same inspection call as mz_pool_read_block: children of the node that owns `block` of `tree`. */
int mzo_pool_read_block(mzo_pool_t* p, int32_t tree, int32_t block, int32_t* out_num_children, int32_t* actions,
                        int32_t* count, double* value_sum, double* prior, float* reward, int32_t* child) {
    const tree_t* tr = &p->trees[tree];
    for (int32_t i = 0; i < tr->num_nodes; ++i) {
        const node_t* nd = &tr->nodes[i];
        if (nd->block != block) continue;
        if (out_num_children) *out_num_children = nd->num_children;
        for (int k = 0; k < nd->num_children; ++k) {
            const node_t* c = &tr->nodes[nd->first_child + k];
            if (actions) actions[k] = c->action;
            if (count) count[k] = c->visit_count;
            if (value_sum) value_sum[k] = c->value_sum;
            if (prior) prior[k] = c->prior;
            if (reward) reward[k] = (float)c->reward;
            if (child) child[k] = c->block;
        }
        return 0;
    }
    return -1;
}
