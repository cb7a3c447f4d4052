/* Compiled (not run) by tests/test_host_logic.py with plain gcc -std=c99: include/qtomo.h must be a C
 * header -- no C++ types, no torch types -- and every entry point must be callable with plain pointers. */
#include <stddef.h>
#include <stdint.h>

#include "../../include/qtomo.h"

int abi_header_check(void) {
  qt_handle_t* h = qt_create(0, 3);
  int64_t counts[27 * 8] = {0};
  double rho[8 * 8 * 2];
  int32_t status = 0;
  int rc;
  if (!h) return qt_device_count() < 0 ? -1 : 1;
  rc = qt_lin_batch(h, counts, 1, 1, rho, NULL, &status, QT_HOST_PTR);
  rc += qt_mle_batch(h, counts, 1, QT_INIT_LIN, 100, 1e-3, rho, NULL, NULL, NULL, &status, QT_HOST_PTR);
  qt_destroy(h);
  return rc + (qt_last_error() != NULL);
}
