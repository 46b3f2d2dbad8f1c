/* Development tool (not product code): a sampling CPU profiler as an LD_PRELOAD library, for boxes without `perf`.
 *
 *   gcc -O2 -fPIC -shared -o /tmp/libcpu_sampler.so tools/csrc/cpu_sampler.c -ldl
 *   CPU_SAMPLER_OUT=/tmp/samples.txt LD_PRELOAD=/tmp/libcpu_sampler.so gst-launch-1.0 ...
 *
 * ITIMER_PROF fires every CPU_SAMPLER_US (default 250) microseconds of process CPU time on whichever thread is running;
 * the handler stores the thread id and the call chain (backtrace(3), warmed up before the timer starts so that it does
 * not allocate in the handler).  At exit the samples are resolved with dladdr(3) and written as three tables: per
 * thread, by leaf function ("self"), and by any function on the chain ("inclusive") -- what a `perf report` of the
 * streaming thread would show.  Used for profiles/r06_element_host.md (where the host time of `hipbayersrc !
 * hipbayer2rgb ! fakesink` goes).  Symbols need the export table only; static functions show up as library+offset. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <unistd.h>

#define MAX_SAMPLES 400000
#define DEPTH 20

typedef struct
{
  int tid;
  int n;
  void *pc[DEPTH];
} sample;

static sample *g_samples;
static volatile int g_n;
static const char *g_out;

static void
on_prof (int sig, siginfo_t * si, void *uc)
{
  int i = __sync_fetch_and_add (&g_n, 1);

  (void) sig;
  (void) si;
  (void) uc;
  if (i >= MAX_SAMPLES)
    return;
  g_samples[i].tid = (int) syscall (SYS_gettid);
  g_samples[i].n = backtrace (g_samples[i].pc, DEPTH);
}

typedef struct
{
  char name[160];
  long self, incl;
} entry;

static entry *g_tab;
static int g_ntab;

static entry *
lookup (const char *name)
{
  int i;

  for (i = 0; i < g_ntab; i++)
    if (strcmp (g_tab[i].name, name) == 0)
      return &g_tab[i];
  if (g_ntab == 8192)
    return &g_tab[0];
  snprintf (g_tab[g_ntab].name, sizeof g_tab[g_ntab].name, "%s", name);
  return &g_tab[g_ntab++];
}

static void
name_of (void *pc, char *out, size_t len)
{
  Dl_info info;

  if (dladdr (pc, &info) && info.dli_fname) {
    const char *base = strrchr (info.dli_fname, '/');

    base = base ? base + 1 : info.dli_fname;
    if (info.dli_sname)
      snprintf (out, len, "%s (%s)", info.dli_sname, base);
    else
      snprintf (out, len, "%s+0x%lx", base, (unsigned long) ((char *) pc - (char *) info.dli_fbase));
  } else {
    snprintf (out, len, "?%p", pc);
  }
}

static int
by_self (const void *a, const void *b)
{
  return (int) (((const entry *) b)->self - ((const entry *) a)->self);
}

static int
by_incl (const void *a, const void *b)
{
  return (int) (((const entry *) b)->incl - ((const entry *) a)->incl);
}

__attribute__ ((destructor))
     static void finish (void)
{
  struct itimerval off = { {0, 0}, {0, 0} };
  int n, i, k, j;
  FILE *f;
  int tids[64], tid_n[64], ntid = 0;

  if (!g_out || !g_samples)
    return;
  setitimer (ITIMER_PROF, &off, NULL);
  n = g_n < MAX_SAMPLES ? g_n : MAX_SAMPLES;
  f = fopen (g_out, "w");
  if (!f)
    return;
  g_tab = calloc (8192, sizeof (entry));
  for (i = 0; i < n; i++) {
    const sample *s = &g_samples[i];
    entry *seen[DEPTH];
    int nseen = 0;
    char nm[160];

    for (k = 0; k < ntid && tids[k] != s->tid; k++);
    if (k == ntid && ntid < 64) {
      tids[ntid] = s->tid;
      tid_n[ntid++] = 0;
    }
    if (k < 64)
      tid_n[k]++;
    /* frames 0..1 are the handler and the signal trampoline */
    for (k = 2; k < s->n; k++) {
      entry *e;

      name_of (s->pc[k], nm, sizeof nm);
      e = lookup (nm);
      if (k == 2)
        e->self++;
      for (j = 0; j < nseen && seen[j] != e; j++);
      if (j == nseen) {
        seen[nseen++] = e;
        e->incl++;
      }
    }
  }
  fprintf (f, "# %d samples, one per %s us of process CPU time\n", n, getenv ("CPU_SAMPLER_US") ? getenv ("CPU_SAMPLER_US") : "250");
  fprintf (f, "## threads\n");
  for (k = 0; k < ntid; k++)
    fprintf (f, "tid %d: %d samples (%.1f %%)\n", tids[k], tid_n[k], 100.0 * tid_n[k] / (n ? n : 1));
  qsort (g_tab, (size_t) g_ntab, sizeof (entry), by_self);
  fprintf (f, "## self (leaf function), top 45\n");
  for (i = 0; i < g_ntab && i < 45; i++)
    fprintf (f, "%6.2f %%  %s\n", 100.0 * g_tab[i].self / (n ? n : 1), g_tab[i].name);
  qsort (g_tab, (size_t) g_ntab, sizeof (entry), by_incl);
  fprintf (f, "## inclusive (anywhere on the call chain), top 60\n");
  for (i = 0; i < g_ntab && i < 60; i++)
    fprintf (f, "%6.2f %%  %s\n", 100.0 * g_tab[i].incl / (n ? n : 1), g_tab[i].name);
  fclose (f);
}

__attribute__ ((constructor))
     static void start (void)
{
  struct sigaction sa;
  struct itimerval it;
  void *warm[4];
  long us;

  g_out = getenv ("CPU_SAMPLER_OUT");
  if (!g_out)
    return;
  /* only in the process that was asked for (gst-launch forks a plugin scanner) */
  if (getenv ("CPU_SAMPLER_MATCH")) {
    char exe[512];
    ssize_t len = readlink ("/proc/self/exe", exe, sizeof exe - 1);

    exe[len > 0 ? len : 0] = '\0';
    if (!strstr (exe, getenv ("CPU_SAMPLER_MATCH"))) {
      g_out = NULL;
      return;
    }
  }
  g_samples = calloc (MAX_SAMPLES, sizeof (sample));
  if (!g_samples)
    return;
  (void) backtrace (warm, 4);   /* loads libgcc's unwinder now, not in the handler */
  memset (&sa, 0, sizeof sa);
  sa.sa_sigaction = on_prof;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction (SIGPROF, &sa, NULL);
  us = getenv ("CPU_SAMPLER_US") ? atol (getenv ("CPU_SAMPLER_US")) : 250;
  it.it_interval.tv_sec = 0;
  it.it_interval.tv_usec = us > 0 ? us : 250;
  it.it_value = it.it_interval;
  setitimer (ITIMER_PROF, &it, NULL);
}
