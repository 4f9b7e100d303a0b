/* synth_text.c — multi-threaded writer of the synthetic alignment text that
 * bench.py's end-to-end legs read (measurement tooling, not part of the
 * product library).  One SAM line per alignment record, trimmed as
 * doc/perform.md:122-128 of the reference recommends (SEQ / QUAL '*'):
 *
 *   <q><read id, 9 digits> \t <flag> \t <s><subject, `swidth` digits> \t <pos>
 *   \t 42 \t <len>M \t * \t 0 \t 0 \t * \t * \n
 *
 * Built by __graft_entry__.build() with gcc into tools/native/libwk_synth.so.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct {
    const char* path;
    int64_t lo, hi, offset, bytes;
    const int64_t* read_id;
    const int32_t* flag;
    const int32_t* subject;
    const int32_t* pos;
    const int32_t* alen;
    char qprefix, sprefix;
    int swidth;
    int seqqual; /* bases of SEQ / QUAL per line; 0: '*' */
    int status;
} Job;

static int udigits(uint32_t v) {
    int d = 1;
    while (v >= 10) {
        v /= 10;
        ++d;
    }
    return d;
}

static char* put_fixed(char* p, uint64_t v, int width) {
    for (int i = width - 1; i >= 0; --i) {
        p[i] = (char)('0' + v % 10);
        v /= 10;
    }
    return p + width;
}

static char* put_uint(char* p, uint32_t v) { return put_fixed(p, v, udigits(v)); }

static int64_t line_len(const Job* j, int64_t i) {
    /* q + 9 + \t + flag + \t + s + swidth + \t + pos + \t42\t + len + "M\t*\t0\t0\t*\t*\n" */
    return 1 + 9 + 1 + udigits(j->flag ? (uint32_t)j->flag[i] : 0u) + 1 + 1 + j->swidth + 1 +
           udigits(j->pos ? (uint32_t)j->pos[i] : 1u) + 4 + udigits(j->alen ? (uint32_t)j->alen[i] : 150u) + 12 +
           (j->seqqual ? 2 * (j->seqqual - 1) : 0);
}

static void* size_job(void* arg) {
    Job* j = (Job*)arg;
    int64_t n = 0;
    for (int64_t i = j->lo; i < j->hi; ++i) n += line_len(j, i);
    j->bytes = n;
    return NULL;
}

static void* write_job(void* arg) {
    Job* j = (Job*)arg;
    static const char tail[] = "M\t*\t0\t0\t*\t*\n";
    static const char tail_sq[] = "M\t*\t0\t0\t";
    const size_t cap = 8u << 20;
    char* buf = (char*)malloc(cap + 256 + 2 * (size_t)j->seqqual);
    int fd = open(j->path, O_WRONLY);
    if (!buf || fd < 0) {
        j->status = -1;
        free(buf);
        if (fd >= 0) close(fd);
        return NULL;
    }
    int64_t off = j->offset;
    char* p = buf;
    for (int64_t i = j->lo; i < j->hi; ++i) {
        *p++ = j->qprefix;
        p = put_fixed(p, (uint64_t)j->read_id[i], 9);
        *p++ = '\t';
        p = put_uint(p, j->flag ? (uint32_t)j->flag[i] : 0u);
        *p++ = '\t';
        *p++ = j->sprefix;
        p = put_fixed(p, (uint64_t)j->subject[i], j->swidth);
        *p++ = '\t';
        p = put_uint(p, j->pos ? (uint32_t)j->pos[i] : 1u);
        memcpy(p, "\t42\t", 4);
        p += 4;
        p = put_uint(p, j->alen ? (uint32_t)j->alen[i] : 150u);
        if (!j->seqqual) {
            memcpy(p, tail, 12);
            p += 12;
        } else { /* SEQ and QUAL of `seqqual` characters, varying with the record */
            memcpy(p, tail_sq, 8);
            p += 8;
            uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + 12345u;
            for (int k = 0; k < j->seqqual; ++k) {
                x ^= x << 13;
                x ^= x >> 7;
                x ^= x << 17;
                *p++ = "ACGT"[x & 3];
            }
            *p++ = '\t';
            for (int k = 0; k < j->seqqual; ++k) {
                x ^= x << 13;
                x ^= x >> 7;
                x ^= x << 17;
                *p++ = (char)('#' + (x & 31) + (x >> 60));
            }
            *p++ = '\n';
        }
        if ((size_t)(p - buf) >= cap || i + 1 == j->hi) {
            size_t left = (size_t)(p - buf);
            const char* q = buf;
            while (left) {
                ssize_t w = pwrite(fd, q, left, off);
                if (w <= 0) {
                    j->status = -1;
                    break;
                }
                q += w;
                off += w;
                left -= (size_t)w;
            }
            p = buf;
            if (j->status) break;
        }
    }
    close(fd);
    free(buf);
    return NULL;
}

/* Returns the file size, or -1. */
int64_t wk_synth_sam(const char* path, int64_t n_rec, const int64_t* read_id, char qprefix, const int32_t* flag,
                     const int32_t* subject, char sprefix, int swidth, const int32_t* pos, const int32_t* alen,
                     int n_threads, int seqqual) {
    static const char head[] = "@HD\tVN:1.0\tSO:unsorted\n";
    if (!path || n_rec < 0 || !read_id || !subject || swidth < 1 || swidth > 12 || seqqual < 0 || seqqual > 100000) return -1;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if ((int64_t)n_threads > n_rec / 65536 + 1) n_threads = (int)(n_rec / 65536 + 1);
    Job* jobs = (Job*)calloc((size_t)n_threads, sizeof(Job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    if (!jobs || !th) return -1;
    for (int t = 0; t < n_threads; ++t) {
        Job* j = &jobs[t];
        j->path = path;
        j->lo = n_rec * t / n_threads;
        j->hi = n_rec * (t + 1) / n_threads;
        j->read_id = read_id;
        j->flag = flag;
        j->subject = subject;
        j->pos = pos;
        j->alen = alen;
        j->qprefix = qprefix;
        j->sprefix = sprefix;
        j->swidth = swidth;
        j->seqqual = seqqual;
    }
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, size_job, &jobs[t]);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    int64_t off = (int64_t)sizeof head - 1;
    for (int t = 0; t < n_threads; ++t) {
        jobs[t].offset = off;
        off += jobs[t].bytes;
    }
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return -1;
    int ok = pwrite(fd, head, sizeof head - 1, 0) == (ssize_t)(sizeof head - 1) && ftruncate(fd, off) == 0;
    close(fd);
    if (!ok) return -1;
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, write_job, &jobs[t]);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    for (int t = 0; t < n_threads; ++t)
        if (jobs[t].status) off = -1;
    free(jobs);
    free(th);
    return off;
}
