/* Host-sanitizer driver (tools/host_sanitize.sh): loads hg_model / hg_config blobs that tools/sanitize/dump_models.py
 * wrote (the structs are plain data), then
 *   - steps the CPU oracle (built with -fsanitize=address,undefined) with random targets, every output array live;
 *   - calls hg_create of the product library (host code built with the same sanitizers) so that its host-side
 *     flattening / validation runs under them; without a GPU it must end in HG_ENODEVICE.
 * Test infrastructure only.  No GPU code runs under a sanitizer (not available on this pool). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hg_model.h"
#include "hg_sim.h"

int hgo_step_full(const hg_model *, const hg_config *, int, double *, const double *, const int32_t *, int, int, double *,
                  double *, double *, uint8_t *, uint64_t *, int32_t *, int32_t *, const double *, double *);
void hgo_reset(const hg_model *, double *, int);
void hgo_link_states(const hg_model *, const double *, int, double *, double *, double *, double *);
void hgo_contact_slack(const hg_model *, const hg_config *, const double *, double *);
void hgo_energy_momentum(const hg_model *, const hg_config *, const double *, double *);

static unsigned long long rng_state = 88172645463325252ull;
static double uni(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

int main(int argc, char **argv) {
    int rc = 0;
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { perror(argv[a]); return 2; }
        hg_model *m = malloc(sizeof *m);
        hg_config *c = malloc(sizeof *c);
        if (fread(m, sizeof *m, 1, f) != 1 || fread(c, sizeof *c, 1, f) != 1) { fprintf(stderr, "%s: short file\n", argv[a]); return 2; }
        fclose(f);
        const int n = m->n_links, S = HG_STATE_SIZE(n), E = 5, na = m->n_act;
        double *state = malloc(sizeof(double) * S * E), *act = malloc(sizeof(double) * E * na);
        double *oq = malloc(sizeof(double) * E * na), *oqd = malloc(sizeof(double) * E * na), *xy = malloc(sizeof(double) * E * 3);
        double *tau = malloc(sizeof(double) * E * n), *ls = malloc(sizeof(double) * E * (n + 1) * 14);
        uint8_t *col = malloc(E);
        uint64_t *mask = malloc(sizeof(uint64_t) * E);
        int32_t *pairs = malloc(sizeof(int32_t) * E * HG_MAX_CONTACTS * 2), *nc = malloc(sizeof(int32_t) * E);
        hgo_reset(m, state, E);
        int contacts = 0;
        for (int t = 0; t < 120; t++) {
            if (t % 20 == 0)
                for (int k = 0; k < E * na; k++) {
                    const int j = m->act_joint[k % na];
                    act[k] = m->lower[j] + uni() * (m->upper[j] - m->lower[j]);
                }
            const int e = hgo_step_full(m, c, E, state, act, NULL, 0, 1 + (t & 1), oq, oqd, xy, col, mask,
                                        pairs, nc, NULL, tau);
            if (e) { fprintf(stderr, "%s: hgo_step_full -> %d at step %d\n", argv[a], e, t); rc = 1; break; }
            for (int k = 0; k < E; k++) contacts += nc[k];
        }
        double slack[2], em[8];
        hgo_link_states(m, state, E, ls, ls + 3 * E * (n + 1), ls + 7 * E * (n + 1), ls + 10 * E * (n + 1));
        hgo_contact_slack(m, c, state, slack);
        hgo_energy_momentum(m, c, state, em);
        hg_sim *sim = NULL;
        const int cr = hg_create(m, c, 4, 0, &sim);
        printf("%s: n=%d oracle 120 steps ok (contacts seen %d, q[0]=%.6f), hg_create -> %d (%s)\n", argv[a], n, contacts,
               state[HG_STATE_BASE], cr, cr ? hg_last_error() : "created");
        if (cr == 0) hg_destroy(sim);
        else if (cr != HG_ENODEVICE) rc = 1;       /* the models are valid: only the missing device may stop it */
        free(state); free(act); free(oq); free(oqd); free(xy); free(tau); free(ls); free(col); free(mask); free(pairs); free(nc);
        free(m); free(c);
    }
    return rc;
}
