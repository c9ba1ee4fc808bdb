/* power_watch.c -- kernel-development aid: sample the GPU's power and shader clock from sysfs (amdgpu hwmon) while something else
 * runs, to see whether a phase sits at the board's power cap.  No ROCm libraries: plain file reads.
 *
 *   gcc -std=c99 -O2 -Wall -Wextra -Werror -pedantic tools/power_watch.c -o /tmp/power_watch
 *   /tmp/power_watch 4000 5 > gpurun_out/power.csv &      # 4000 ms, one sample every 5 ms: t_ms, watts, cap_watts, sclk_mhz, temp_c
 *   /tmp/c_bench --steps 4 --warmup 1 ; wait
 *
 * It prints which files it found on stderr and exits 3 if the box exposes none (containers often hide hwmon).  The question it is
 * for ("where round 5 starts"): do the NAR stages (gemm_f16x2, ~230 ms of every 700 ms batch) run AT the cap? */
#define _POSIX_C_SOURCE 200809L
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static int find(const char* leaf, char* out, size_t n) {
  for (int card = 0; card < 16; ++card) {
    char base[256];
    snprintf(base, sizeof base, "/sys/class/drm/card%d/device/hwmon", card);
    DIR* d = opendir(base);
    if (!d) continue;
    struct dirent* e;
    while ((e = readdir(d)) != NULL) {
      if (strncmp(e->d_name, "hwmon", 5)) continue;
      if (strlen(base) + strlen(e->d_name) + strlen(leaf) + 3 > n) continue;
      strcpy(out, base); strcat(out, "/"); strcat(out, e->d_name); strcat(out, "/"); strcat(out, leaf);
      FILE* f = fopen(out, "r");
      if (f) { fclose(f); closedir(d); return 1; }
    }
    closedir(d);
  }
  out[0] = 0;
  return 0;
}

static double read_num(const char* path) {
  if (!path[0]) return -1.0;
  FILE* f = fopen(path, "r");
  if (!f) return -1.0;
  double v = -1.0;
  if (fscanf(f, "%lf", &v) != 1) v = -1.0;
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  const int total_ms = argc > 1 ? atoi(argv[1]) : 3000, every_ms = argc > 2 ? atoi(argv[2]) : 5;
  char p_avg[512], p_in[512], p_cap[512], f_sclk[512], t_edge[512];
  const int have_avg = find("power1_average", p_avg, sizeof p_avg), have_in = find("power1_input", p_in, sizeof p_in);
  find("power1_cap", p_cap, sizeof p_cap);
  find("freq1_input", f_sclk, sizeof f_sclk);
  find("temp1_input", t_edge, sizeof t_edge);
  fprintf(stderr, "[power_watch] power1_average: %s\n[power_watch] power1_input: %s\n[power_watch] power1_cap: %s\n"
                  "[power_watch] freq1_input: %s\n[power_watch] temp1_input: %s\n",
          p_avg[0] ? p_avg : "-", p_in[0] ? p_in : "-", p_cap[0] ? p_cap : "-", f_sclk[0] ? f_sclk : "-", t_edge[0] ? t_edge : "-");
  if (!have_avg && !have_in) { fprintf(stderr, "[power_watch] no amdgpu hwmon power file visible on this box\n"); return 3; }
  const char* pw = have_in ? p_in : p_avg;            /* instantaneous if the driver offers it, else its running average */
  const double cap = read_num(p_cap);
  struct timespec t0, t;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  printf("t_ms,watts,cap_watts,sclk_mhz,temp_c\n");
  for (;;) {
    clock_gettime(CLOCK_MONOTONIC, &t);
    const double ms = (t.tv_sec - t0.tv_sec) * 1e3 + (t.tv_nsec - t0.tv_nsec) * 1e-6;
    if (ms > total_ms) break;
    const double w = read_num(pw), hz = read_num(f_sclk), mc = read_num(t_edge);
    printf("%.1f,%.1f,%.1f,%.0f,%.1f\n", ms, w < 0 ? -1.0 : w * 1e-6, cap < 0 ? -1.0 : cap * 1e-6, hz < 0 ? -1.0 : hz * 1e-6,
           mc < 0 ? -1.0 : mc * 1e-3);
    struct timespec nap = {0, (long)every_ms * 1000000L};
    nanosleep(&nap, NULL);
  }
  return 0;
}
