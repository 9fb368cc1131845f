/*
 * dsm_safetensors.h — minimal read-only safetensors reader (plain C, header-only).
 * Format: u64 LE header length N, N bytes of JSON {"name":{"dtype":"F32","shape":[..],
 * "data_offsets":[begin,end]}, ..., "__metadata__":{...}}, then the raw little-endian data.
 * Replaces candle::safetensors::load / VarBuilder::from_mmaped_safetensors
 * (reference: srv/batched_asr.rs:738-753, core/mimi.rs:261-276).  Executes nothing from the
 * file; every offset is bounds-checked against the mapping.
 */
#ifndef DSM_SAFETENSORS_H
#define DSM_SAFETENSORS_H

#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define DSM_ST_MAX_DIMS 8

typedef enum { DSM_ST_F32 = 0, DSM_ST_BF16 = 1, DSM_ST_F16 = 2, DSM_ST_OTHER = 3 } dsm_st_dtype;

typedef struct dsm_st_tensor {
  char* name;
  dsm_st_dtype dtype;
  int ndim;
  int64_t shape[DSM_ST_MAX_DIMS];
  const uint8_t* data; /* into the mapping */
  size_t nbytes;
} dsm_st_tensor;

typedef struct dsm_st_file {
  int fd;
  uint8_t* map;
  size_t map_size;
  dsm_st_tensor* tensors;
  int n_tensors;
  char err[256];
} dsm_st_file;

static inline void dsm_st__skip_ws(const char** p, const char* end) {
  while (*p < end && (**p == ' ' || **p == '\n' || **p == '\t' || **p == '\r')) ++*p;
}

/* Parses a JSON string at *p (which must point at '"'); returns malloc'd copy (escapes kept verbatim except \" and \\). */
static inline char* dsm_st__parse_string(const char** p, const char* end) {
  if (*p >= end || **p != '"') return NULL;
  ++*p;
  const char* s = *p;
  size_t cap = 64, n = 0;
  char* out = (char*)malloc(cap);
  while (s < end && *s != '"') {
    char c = *s;
    if (c == '\\' && s + 1 < end) {
      ++s;
      c = *s;
    }
    if (n + 2 > cap) {
      cap *= 2;
      out = (char*)realloc(out, cap);
    }
    out[n++] = c;
    ++s;
  }
  if (s >= end) {
    free(out);
    return NULL;
  }
  out[n] = 0;
  *p = s + 1;
  return out;
}

/* Skips any JSON value (used for __metadata__). */
static inline int dsm_st__skip_value(const char** p, const char* end) {
  dsm_st__skip_ws(p, end);
  if (*p >= end) return -1;
  char c = **p;
  if (c == '"') {
    char* s = dsm_st__parse_string(p, end);
    if (!s) return -1;
    free(s);
    return 0;
  }
  if (c == '{' || c == '[') {
    char open = c, close = (c == '{') ? '}' : ']';
    int depth = 0;
    while (*p < end) {
      char d = **p;
      if (d == '"') {
        char* s = dsm_st__parse_string(p, end);
        if (!s) return -1;
        free(s);
        continue;
      }
      if (d == open) depth++;
      if (d == close) {
        depth--;
        if (depth == 0) {
          ++*p;
          return 0;
        }
      }
      ++*p;
    }
    return -1;
  }
  while (*p < end && **p != ',' && **p != '}' && **p != ']') ++*p;
  return 0;
}

static inline int dsm_st__parse_int(const char** p, const char* end, int64_t* out) {
  dsm_st__skip_ws(p, end);
  int64_t v = 0;
  int any = 0;
  while (*p < end && **p >= '0' && **p <= '9') {
    if (v > (INT64_MAX - 9) / 10) return -1; /* a hostile header must not overflow the accumulator */
    v = v * 10 + (**p - '0');
    ++*p;
    any = 1;
  }
  *out = v;
  return any ? 0 : -1;
}

static inline void dsm_st_close(dsm_st_file* f) {
  if (!f) return;
  for (int i = 0; i < f->n_tensors; ++i) free(f->tensors[i].name);
  free(f->tensors);
  if (f->map) munmap(f->map, f->map_size);
  if (f->fd >= 0) close(f->fd);
  free(f);
}

static inline dsm_st_file* dsm_st_open(const char* path, char* err, size_t errcap) {
  dsm_st_file* f = (dsm_st_file*)calloc(1, sizeof(dsm_st_file));
  f->fd = open(path, O_RDONLY);
  if (f->fd < 0) {
    snprintf(err, errcap, "cannot open %s", path);
    free(f);
    return NULL;
  }
  struct stat st;
  if (fstat(f->fd, &st) != 0 || st.st_size < 8) {
    snprintf(err, errcap, "%s: too small for a safetensors file", path);
    close(f->fd);
    free(f);
    return NULL;
  }
  f->map_size = (size_t)st.st_size;
  f->map = (uint8_t*)mmap(NULL, f->map_size, PROT_READ, MAP_PRIVATE, f->fd, 0);
  if (f->map == MAP_FAILED) {
    snprintf(err, errcap, "%s: mmap failed", path);
    close(f->fd);
    free(f);
    return NULL;
  }
  uint64_t hlen;
  memcpy(&hlen, f->map, 8);
  if (hlen > f->map_size - 8) {
    snprintf(err, errcap, "%s: header length %llu exceeds file", path, (unsigned long long)hlen);
    dsm_st_close(f);
    return NULL;
  }
  const char* p = (const char*)f->map + 8;
  const char* end = p + hlen;
  const uint8_t* data_base = f->map + 8 + hlen;
  size_t data_size = f->map_size - 8 - hlen;
  int cap = 256;
  f->tensors = (dsm_st_tensor*)calloc(cap, sizeof(dsm_st_tensor));
  dsm_st__skip_ws(&p, end);
  if (p >= end || *p != '{') goto bad;
  ++p;
  for (;;) {
    dsm_st__skip_ws(&p, end);
    if (p < end && *p == '}') break;
    char* name = dsm_st__parse_string(&p, end);
    if (!name) goto bad;
    dsm_st__skip_ws(&p, end);
    if (p >= end || *p != ':') {
      free(name);
      goto bad;
    }
    ++p;
    if (strcmp(name, "__metadata__") == 0) {
      free(name);
      if (dsm_st__skip_value(&p, end)) goto bad;
    } else {
      dsm_st_tensor t;
      memset(&t, 0, sizeof t);
      t.name = name;
      t.dtype = DSM_ST_OTHER;
      int64_t off0 = -1, off1 = -1;
      dsm_st__skip_ws(&p, end);
      if (p >= end || *p != '{') {
        free(name);
        goto bad;
      }
      ++p;
      for (;;) {
        dsm_st__skip_ws(&p, end);
        if (p < end && *p == '}') {
          ++p;
          break;
        }
        char* key = dsm_st__parse_string(&p, end);
        if (!key) {
          free(name);
          goto bad;
        }
        dsm_st__skip_ws(&p, end);
        if (p < end && *p == ':') ++p;
        dsm_st__skip_ws(&p, end);
        if (strcmp(key, "dtype") == 0) {
          char* v = dsm_st__parse_string(&p, end);
          if (v) {
            if (!strcmp(v, "F32")) t.dtype = DSM_ST_F32;
            else if (!strcmp(v, "BF16")) t.dtype = DSM_ST_BF16;
            else if (!strcmp(v, "F16")) t.dtype = DSM_ST_F16;
            free(v);
          }
        } else if (strcmp(key, "shape") == 0 || strcmp(key, "data_offsets") == 0) {
          int is_shape = key[0] == 's';
          if (p < end && *p == '[') ++p;
          int n = 0;
          for (;;) {
            dsm_st__skip_ws(&p, end);
            if (p < end && *p == ']') {
              ++p;
              break;
            }
            int64_t v;
            if (dsm_st__parse_int(&p, end, &v)) {
              free(key);
              free(name);
              goto bad;
            }
            if (is_shape) {
              if (n < DSM_ST_MAX_DIMS) t.shape[n] = v;
            } else {
              if (n == 0) off0 = v;
              if (n == 1) off1 = v;
            }
            ++n;
            dsm_st__skip_ws(&p, end);
            if (p < end && *p == ',') ++p;
          }
          if (is_shape) t.ndim = n < DSM_ST_MAX_DIMS ? n : DSM_ST_MAX_DIMS;
        } else {
          if (dsm_st__skip_value(&p, end)) {
            free(key);
            free(name);
            goto bad;
          }
        }
        free(key);
        dsm_st__skip_ws(&p, end);
        if (p < end && *p == ',') ++p;
      }
      if (off0 < 0 || off1 < off0 || (uint64_t)off1 > data_size) {
        snprintf(err, errcap, "%s: tensor %s has data_offsets outside the file", path, name);
        free(name);
        dsm_st_close(f);
        return NULL;
      }
      t.data = data_base + off0;
      t.nbytes = (size_t)(off1 - off0);
      if (f->n_tensors == cap) {
        cap *= 2;
        f->tensors = (dsm_st_tensor*)realloc(f->tensors, cap * sizeof(dsm_st_tensor));
      }
      f->tensors[f->n_tensors++] = t;
    }
    dsm_st__skip_ws(&p, end);
    if (p < end && *p == ',') ++p;
  }
  return f;
bad:
  snprintf(err, errcap, "%s: malformed safetensors header", path);
  dsm_st_close(f);
  return NULL;
}

static inline const dsm_st_tensor* dsm_st_find(const dsm_st_file* f, const char* name) {
  for (int i = 0; i < f->n_tensors; ++i)
    if (strcmp(f->tensors[i].name, name) == 0) return &f->tensors[i];
  return NULL;
}

static inline int64_t dsm_st_numel(const dsm_st_tensor* t) {
  int64_t n = 1;
  for (int i = 0; i < t->ndim; ++i) {
    if (t->shape[i] < 0 || (t->shape[i] > 0 && n > INT64_MAX / t->shape[i])) return -1; /* overflow: matches no request */
    n *= t->shape[i];
  }
  return n;
}

/* Copies tensor `name` as f32 (upcasting BF16 exactly; F16 unsupported here) after checking
 * that it holds exactly `numel` elements.  Returns 0 or -1 with err filled. */
static inline int dsm_st_read_f32(const dsm_st_file* f, const char* name, int64_t numel, float* out,
                                  char* err, size_t errcap) {
  const dsm_st_tensor* t = dsm_st_find(f, name);
  if (!t) {
    snprintf(err, errcap, "cannot find tensor %s", name);
    return -1;
  }
  int64_t n = dsm_st_numel(t);
  if (n != numel) {
    snprintf(err, errcap, "shape mismatch for %s: file has %lld elements, expected %lld", name, (long long)n,
             (long long)numel);
    return -1;
  }
  if (t->dtype == DSM_ST_F32) {
    if (t->nbytes != (size_t)n * 4) goto size;
    memcpy(out, t->data, (size_t)n * 4);
    return 0;
  }
  if (t->dtype == DSM_ST_BF16) {
    if (t->nbytes != (size_t)n * 2) goto size;
    const uint16_t* s = (const uint16_t*)t->data;
    for (int64_t i = 0; i < n; ++i) {
      uint32_t u = ((uint32_t)s[i]) << 16;
      memcpy(&out[i], &u, 4);
    }
    return 0;
  }
  snprintf(err, errcap, "unsupported dtype for %s (need F32 or BF16)", name);
  return -1;
size:
  snprintf(err, errcap, "byte size of %s does not match its shape", name);
  return -1;
}

#endif /* DSM_SAFETENSORS_H */
