/* TEST SCAFFOLDING (see ../mi355_gst_stub.h): stands in for <gst/base/gstaggregator.h> under `make -C gst syntax`. */
#include "../../mi355_gst_stub.h"
