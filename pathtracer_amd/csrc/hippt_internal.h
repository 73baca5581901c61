// hippt_internal.h — shared between the translation units of libhippt.
#pragma once
#include "../../include/hippt.h"

// Stores a printf-style message for hippt_last_error() and returns `code`.
int hippt_set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
