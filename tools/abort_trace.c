/* LD_PRELOAD helper for hunting a silent abort() in a long GPU test run: on SIGABRT / SIGSEGV / SIGBUS it writes the C backtrace of
 * the thread that raised it (the Python fault handler only shows threads that hold Python frames) and /proc/self/maps (to resolve the
 * frames with addr2line afterwards) to stderr and to $ABORT_TRACE_FILE, then lets the signal take its default course.
 *   gcc -O1 -g -shared -fPIC -o abort_trace.so abort_trace.c
 *   LD_PRELOAD=tools/abort_trace.so ABORT_TRACE_FILE=gpurun_out/abort_bt.txt python -m pytest -s -p no:faulthandler ...            */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <sys/wait.h>
#include <unistd.h>

static void put(int fd, const char *s) { if (write(fd, s, strlen(s)) < 0) {} }

static void dump(int fd, int sig)
{
	char line[128];
	char name[32] = "?";
	int nfd = open("/proc/thread-self/comm", O_RDONLY);
	if (nfd >= 0) {
		ssize_t r = read(nfd, name, sizeof name - 1);
		if (r > 0) name[r - 1] = 0;
		close(nfd);
	}
	snprintf(line, sizeof line, "\n=== abort_trace: signal %d on tid %ld (%s), pid %d\n", sig, (long)syscall(SYS_gettid), name, (int)getpid());
	put(fd, line);
	void *frames[96];
	int n = backtrace(frames, 96);
	backtrace_symbols_fd(frames, n, fd);
	put(fd, "=== maps (executable ranges)\n");
	int m = open("/proc/self/maps", O_RDONLY);
	if (m >= 0) {
		static char buf[1 << 20];
		ssize_t got, len = 0;
		while ((got = read(m, buf + len, sizeof buf - 1 - len)) > 0) len += got;
		buf[len] = 0;
		close(m);
		for (char *p = buf; *p;) {
			char *e = strchr(p, '\n');
			if (!e) e = p + strlen(p);
			char save = *e;
			*e = 0;
			if (strstr(p, " r-xp ")) { put(fd, p); put(fd, "\n"); }
			*e = save;
			p = save ? e + 1 : e;
		}
	}
	put(fd, "=== end abort_trace\n");
}

/* ABORT_TRACE_GDB=<file>: before the process dies, rocgdb attaches to it and lists the device's queues, dispatches and waves
 * (a queue exception leaves the faulting waves halted: "info threads" names the kernel they are in). */
static void gdb_snapshot(void)
{
	const char *out = getenv("ABORT_TRACE_GDB");
	if (!out || !*out)
		return;
	char pid[32];
	snprintf(pid, sizeof pid, "%d", (int)getpid());
	pid_t child = fork();
	if (child == 0) {
		int fd = open(out, O_WRONLY | O_CREAT | O_APPEND, 0644);
		if (fd >= 0) { dup2(fd, 1); dup2(fd, 2); }
		unsetenv("LD_PRELOAD");
		alarm(240);
		execl("/opt/rocm/bin/rocgdb", "rocgdb", "-q", "-batch", "-p", pid, "-ex", "set pagination off", "-ex", "info agents", "-ex", "info queues",
		      "-ex", "info dispatches", "-ex", "info threads", "-ex", "thread apply all bt 14", (char *)NULL);
		_exit(127);
	}
	if (child > 0) {
		int st;
		waitpid(child, &st, 0);
	}
}

static void handler(int sig)
{
	dump(2, sig);
	const char *path = getenv("ABORT_TRACE_FILE");
	if (path) {
		int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
		if (fd >= 0) { dump(fd, sig); close(fd); }
	}
	if (sig == SIGABRT)
		gdb_snapshot();
	signal(sig, SIG_DFL);
	raise(sig);
}

__attribute__((constructor)) static void install(void)
{
	void *warm[4];
	backtrace(warm, 4);   /* loads libgcc now, not inside the handler */
	struct sigaction sa;
	memset(&sa, 0, sizeof sa);
	sa.sa_handler = handler;
	sa.sa_flags = SA_NODEFER;
	sigaction(SIGABRT, &sa, NULL);
	sigaction(SIGSEGV, &sa, NULL);
	sigaction(SIGBUS, &sa, NULL);
}
