#include <stdio.h>
#include <string.h>
#include "pb_oracle.h"
int main(int argc, char **argv) {
  const char *names[] = {"example.cfg", "example_dead_cells.cfg", "example_gap.cfg", "example_object_transport.cfg", "example_obstacle.cfg"};
  for (int k = 0; k < 5; k++) {
    char path[512];
    snprintf(path, sizeof path, "%s/examples/%s", argv[1], names[k]);
    OrcParams P;
    orc_params_defaults(&P);
    if (orc_load_cfg(&P, path)) return 1;
    P.time_to_dead = 0.05f; P.light_shadow = (k % 3);
    P.rngKind = k % 3; /* counter generator, cuRAND-compatible XORWOW, rocRAND-seeded XORWOW */
    orc_params_derive(&P, 0, 0.0f);
    OrcSim *s = orc_sim_create(&P);
    orc_sim_reset(s, 0);
    for (int i = 0; i < 1300; i++) orc_sim_update(s, 0.01f, 0.7f);
    FILE *fp = fopen("o.csv", "w+");
    orc_sim_dump(s, fp, 1.0f, 1, 0);
    rewind(fp);
    fclose(fp);
    orc_sim_destroy(s);
  }
  {
    uint32_t u[32], rows[800];
    float z[3 * 50];
    orc_xorwow_outputs(1, 0xFFFFFFFFFFFFull, 70000u, 32, u);
    orc_xorwow_normals(2, 9u, 50, 3, z);
    orc_xorwow_jump_rows(rows);
  }
  printf("oracle sanitize done\n");
  return 0;
}
