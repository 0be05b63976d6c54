extern "C" const char* uz_source_hash(void) { return "a8ab79dacf54b6908ad53675e8a4bae015c9d3ea84f6e97a882ae1857563506b"; }
