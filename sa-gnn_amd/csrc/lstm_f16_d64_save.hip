// Split-f16 interval LSTM, d = 64, training forward (stores gate activations and cell states).
#include "lstm_f16_kernel.h"

namespace sagnn {
int lstm_f16_d64_save(SAGNN_LSTM_F16_ARGS) { return launch_lstm_f16<64, true, false>(SAGNN_LSTM_F16_PASS); }
}  // namespace sagnn
